/*
 * sonic_oracle.c -- CPU restatement of the SonicScribe hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP engine.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (sonicscribe_amd/) never does.
 *
 * What it restates (HF = transformers 5.15.0, the third-party library all of
 * backend/asr.py's arithmetic lives in -- pinned "git main" in backend/requirements.txt:12):
 *   a5  log-mel front-end        HF:models/whisper/feature_extraction_whisper.py:95-103,135-168,300-341
 *                                HF:audio_utils.py:448-518 (slaney mel scale), 541-560, 690-722
 *   a7  conv stem + GELU         HF:models/glmasr/modeling_glmasr.py:313-316
 *   a8  encoder layers           modeling_glmasr.py:50-106 (rotary), 153-168 (partial rope),
 *                                171-221 (attention, k_proj without bias), 224-270, 319-327
 *   a9  4-frame merge + projector modeling_glmasr.py:330-346, 380-408
 *   a10 embedding + audio scatter modeling_glmasr.py:452-465
 *   a11 Llama decoder            HF:models/llama/modeling_llama.py:53-67 (RMSNorm), 73-160 (rope),
 *                                163-176 (SwiGLU), 217-324; SDPA semantics HF:integrations/sdpa_attention.py
 *   a12 lm_head + greedy loop    modeling_glmasr.py:584-586; HF:generation/utils.py:2876-2943
 *
 * Pinning: validated in the build container against the reference arithmetic itself (HF modules
 * imported by oracle/gen_golden.py, which writes tests/golden/ npz files); tests/test_oracle_golden.py
 * replays those fixtures through this file.  The reference tree holds no tests or golden vectors of
 * its own, and no real checkpoint exists offline: real-weight transcripts are unpinned.
 *
 * Two numeric modes:
 *   fp32  (bf16 = 0): every value fp32, as HF runs with dtype=float32.
 *   bf16  (bf16 = 1): values rounded to bfloat16 (RNE) at every torch op boundary, reproducing
 *         the reference's `mode="native"` (torch.bfloat16 weights and activations, fp32 accumulation
 *         inside each op) -- asr.py:61,280-301.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#pragma STDC FP_CONTRACT OFF

#define N_FFT 400
#define HOP 160
#define N_BINS 201

typedef struct {
    int n_mels, n_frames, enc_T;
    int enc_d, enc_ff, enc_layers, enc_heads, enc_rotary_dim;
    float enc_theta, enc_ln_eps;
    int merge;
    int dec_d, dec_ff, dec_layers, dec_heads, dec_kv_heads, dec_head_dim;
    float dec_theta, dec_rms_eps;
    int vocab, audio_token_id, n_eos;
    int eos[8];
} oracle_dims;

typedef struct {
    float *conv1, *conv2;      /* pre-GELU conv outputs [C][n_frames], [C][enc_T] (HF hook layout) */
    float *enc_layers;         /* [L][T][d] */
    float *enc_out;            /* [T][d] after the final LayerNorm */
    float *audio_embeds;       /* [n_audio][dec_d] */
    float *dec_layers;         /* [L][P][dec_d] at prefill */
    float *prefill_logits;     /* [vocab] last prompt position */
    float *step_logits;        /* [max_new][vocab]; row 0 == prefill_logits */
    int *new_ids;              /* [max_new] */
    int *n_new;
    const int *force_ids;      /* optional teacher forcing: feed these instead of the argmax */
} oracle_outputs;

typedef struct {
    oracle_dims d;
    int bf16;
    float **w; /* tensor pointers in spec.tensor_inventory order (borrowed) */
} oracle_model;

/* ---------------------------------------------------------------- bf16 + synth */
static inline float bf16_round(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&x, &u, 4); return x;
}
#define RB(m, x) ((m)->bf16 ? bf16_round(x) : (x))

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static uint64_t fnv1a64(const char *s) {
    uint64_t h = 0xCBF29CE484222325ULL;
    for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ULL; }
    return h;
}
/* sonicscribe_amd/synth.py: the portable generator */
void oracle_synth_fill(uint64_t seed, const char *name, long n, float scale, float offset, int bf16, float *out) {
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    uint64_t key = mix64(seed * G + fnv1a64(name));
    #pragma omp parallel for schedule(static) if (n > (1L << 16))
    for (long i = 0; i < n; ++i) {
        uint64_t z = mix64(key + (uint64_t)(i + 1) * G);
        int32_t bits = (int32_t)(z >> 40);
        volatile float r = (float)bits * 0x1p-23f - 1.0f;
        volatile float p = r * scale;
        volatile float v = offset + p;
        out[i] = bf16 ? bf16_round(v) : v;
    }
}

/* ---------------------------------------------------------------- a5: log-mel */
static double hz_to_mel(double f) { /* audio_utils.py:448-481, slaney */
    if (f >= 1000.0) return 15.0 + log(f / 1000.0) * (27.0 / log(6.4));
    return 3.0 * f / 200.0;
}
static double mel_to_hz(double m) { /* audio_utils.py:484-518 */
    if (m >= 15.0) return 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0));
    return 200.0 * m / 3.0;
}
/* mel_filter_bank(201, n_mels, 0, 8000, 16000, "slaney", "slaney") -> [201][n_mels] f64 -> f32 */
void oracle_mel_filters(int n_mels, float *out /* [N_BINS][n_mels] */) {
    int nf = n_mels + 2;
    double *ff = (double *)malloc(sizeof(double) * nf);
    double mmin = hz_to_mel(0.0), mmax = hz_to_mel(8000.0);
    for (int i = 0; i < nf; ++i) { /* np.linspace */
        double step = (mmax - mmin) / (nf - 1);
        double m = (i == nf - 1) ? mmax : mmin + step * i;
        ff[i] = mel_to_hz(m);
    }
    for (int b = 0; b < N_BINS; ++b) {
        double step = 8000.0 / (N_BINS - 1);
        double fb = (b == N_BINS - 1) ? 8000.0 : step * b;
        for (int m = 0; m < n_mels; ++m) {
            double down = -(ff[m] - fb) / (ff[m + 1] - ff[m]);
            double up = (ff[m + 2] - fb) / (ff[m + 2] - ff[m + 1]);
            double v = down < up ? down : up;
            if (v < 0) v = 0;
            v *= 2.0 / (ff[m + 2] - ff[m]);
            out[b * n_mels + m] = (float)v;
        }
    }
    free(ff);
}

/* pcm: int16 after the asr.py:247-276 normalise + PCM_16 round trip; HF load_audio gives s/32768.
 * feats: [n_mels][n_frames] fp32 (HF layout), mask: [n_frames]. */
void oracle_logmel(const int16_t *pcm, int n, int n_mels, int n_frames, float *feats, int *mask) {
    const int n_pad = n_frames * HOP; /* 480000 */
    if (n > n_pad) n = n_pad;
    float *filt = (float *)malloc(sizeof(float) * N_BINS * n_mels);
    oracle_mel_filters(n_mels, filt);
    float win[N_FFT];
    for (int i = 0; i < N_FFT; ++i) /* torch.hann_window(400) (periodic), fp32 */
        win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / N_FFT));
    double *ct = (double *)malloc(sizeof(double) * N_FFT), *st = (double *)malloc(sizeof(double) * N_FFT);
    for (int i = 0; i < N_FFT; ++i) { ct[i] = cos(2.0 * M_PI * i / N_FFT); st[i] = sin(2.0 * M_PI * i / N_FFT); }
    float *logspec = (float *)malloc(sizeof(float) * n_mels * n_frames);
    /* last frame that can see a non-zero sample: frame t covers padded [t*160-200, t*160+200) */
    #pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < n_frames; ++t) {
        float fr[N_FFT];
        int any = 0;
        for (int i = 0; i < N_FFT; ++i) {
            int j = t * HOP + i - N_FFT / 2; /* center=True */
            if (j < 0) j = -j;                              /* reflect pad (torch.stft pad_mode="reflect") */
            if (j >= n_pad) j = 2 * (n_pad - 1) - j;
            float x = (j < n) ? (float)pcm[j] / 32768.0f : 0.0f; /* zero pad to 30 s (feature_extraction_whisper.py:300-307) */
            fr[i] = x * win[i];
            any |= (x != 0.0f);
        }
        float power[N_BINS];
        if (!any) { for (int b = 0; b < N_BINS; ++b) power[b] = 0.0f; }
        else for (int b = 0; b < N_BINS; ++b) {
            double re = 0, im = 0;
            for (int i = 0; i < N_FFT; ++i) {
                int k = (b * i) % N_FFT;
                re += fr[i] * ct[k]; im -= fr[i] * st[k];
            }
            float a = hypotf((float)re, (float)im); /* stft.abs() ** 2 (:154) */
            power[b] = a * a;
        }
        for (int m = 0; m < n_mels; ++m) {
            double acc = 0;
            for (int b = 0; b < N_BINS; ++b) acc += (double)filt[b * n_mels + m] * power[b];
            float v = (float)acc;
            if (v < 1e-10f) v = 1e-10f;            /* clamp(min=1e-10).log10() (:159) */
            logspec[m * n_frames + t] = log10f(v);
        }
    }
    float gmax = -1e30f;
    for (long i = 0; i < (long)n_mels * n_frames; ++i) if (logspec[i] > gmax) gmax = logspec[i];
    for (long i = 0; i < (long)n_mels * n_frames; ++i) {
        float v = logspec[i];
        float fl = gmax - 8.0f;                   /* maximum(log_spec, max - 8) (:160-164) */
        if (v < fl) v = fl;
        feats[i] = (v + 4.0f) / 4.0f;
    }
    if (mask) for (int t = 0; t < n_frames; ++t) mask[t] = (t * HOP < n) ? 1 : 0; /* attention_mask[:, ::160] (:332-341) */
    free(filt); free(ct); free(st); free(logspec);
}

/* ---------------------------------------------------------------- dense helpers */
/* y[T][N] = x[T][K] (row stride ldx) @ w[N][K]^T + b ; optional rounding */
static void linear(const oracle_model *m, const float *x, long ldx, const float *w, const float *b,
                   float *y, long ldy, int T, int N, int K) {
    #pragma omp parallel for schedule(static)
    for (int t0 = 0; t0 < T; t0 += 4) {
        int tn = T - t0 < 4 ? T - t0 : 4;
        for (int o = 0; o < N; ++o) {
            const float *wr = w + (long)o * K;
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            const float *x0 = x + (long)t0 * ldx, *x1 = x0 + (tn > 1 ? ldx : 0), *x2 = x0 + (tn > 2 ? 2 * ldx : 0), *x3 = x0 + (tn > 3 ? 3 * ldx : 0);
            #pragma omp simd reduction(+ : a0, a1, a2, a3)
            for (int k = 0; k < K; ++k) { float wv = wr[k]; a0 += x0[k] * wv; a1 += x1[k] * wv; a2 += x2[k] * wv; a3 += x3[k] * wv; }
            float bb = b ? b[o] : 0.0f;
            float acc[4] = {a0 + bb, a1 + bb, a2 + bb, a3 + bb};
            for (int i = 0; i < tn; ++i) y[(long)(t0 + i) * ldy + o] = RB(m, acc[i]);
        }
    }
}
static inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
static inline float silu(float x) { return x / (1.0f + expf(-x)); }

static void layernorm(const oracle_model *m, const float *x, const float *w, const float *b, float *y, int T, int d, float eps) {
    #pragma omp parallel for
    for (int t = 0; t < T; ++t) {
        const float *xr = x + (long)t * d; float *yr = y + (long)t * d;
        double s = 0; for (int i = 0; i < d; ++i) s += xr[i];
        double mean = s / d, v = 0;
        for (int i = 0; i < d; ++i) { double c = xr[i] - mean; v += c * c; }
        float rstd = (float)(1.0 / sqrt(v / d + eps));
        for (int i = 0; i < d; ++i) yr[i] = RB(m, ((xr[i] - (float)mean) * rstd) * w[i] + b[i]);
    }
}
/* LlamaRMSNorm (modeling_llama.py:60-65): fp32 normalise, cast to input dtype, then * weight */
static void rmsnorm(const oracle_model *m, const float *x, const float *w, float *y, int T, int d, float eps) {
    #pragma omp parallel for
    for (int t = 0; t < T; ++t) {
        const float *xr = x + (long)t * d; float *yr = y + (long)t * d;
        float s = 0; for (int i = 0; i < d; ++i) s += xr[i] * xr[i];
        float r = 1.0f / sqrtf(s / d + eps);
        for (int i = 0; i < d; ++i) { float h = RB(m, xr[i] * r); yr[i] = RB(m, w[i] * h); }
    }
}
/* rotate-half RoPE on the first rd dims of every head (modeling_glmasr.py:153-168; llama:121-143).
 * cos/sin are computed in fp32 and cast to the activation dtype before use (:104-106). */
static void rope(const oracle_model *m, float *x, int T, int heads, int hd, int rd, float theta, const int *pos) {
    int half = rd / 2;
    #pragma omp parallel for
    for (int t = 0; t < T; ++t) {
        float c[128], s[128];
        for (int i = 0; i < half; ++i) {
            float inv = 1.0f / powf(theta, (float)(2 * i) / (float)rd);
            float ang = inv * (float)pos[t];
            c[i] = RB(m, cosf(ang)); s[i] = RB(m, sinf(ang));
        }
        for (int h = 0; h < heads; ++h) {
            float *v = x + ((long)t * heads + h) * hd;
            for (int i = 0; i < half; ++i) {
                float x1 = v[i], x2 = v[i + half];
                float o1 = RB(m, RB(m, x1 * c[i]) + RB(m, -x2 * s[i]));
                float o2 = RB(m, RB(m, x2 * c[i]) + RB(m, x1 * s[i]));
                v[i] = o1; v[i + half] = o2;
            }
        }
    }
}
/* SDPA: out[t][h] = softmax(q.k * scale) v.  q:[Tq][Hq][hd], k,v:[Tk][Hkv][hd] (row strides given).
 * causal: query t (absolute position q_pos0 + t) sees keys j <= q_pos0 + t.  Probabilities are kept in
 * fp32 for the row sum and rounded to the activation dtype for the P.V product, as torch's CPU
 * flash kernel does for reduced types (aten FlashAttentionKernel: qk fp32 -> exp/sum fp32 -> P as bf16). */
static void attention(const oracle_model *m, const float *q, long ldq, const float *k, const float *v, long ldkv,
                      float *out, long ldo, int Tq, int Tk, int Hq, int Hkv, int hd, int causal, int q_pos0) {
    float scale = 1.0f / sqrtf((float)hd);
    int grp = Hq / Hkv;
    #pragma omp parallel
    {
        float *s = (float *)malloc(sizeof(float) * Tk);
        #pragma omp for collapse(2) schedule(dynamic, 8)
        for (int h = 0; h < Hq; ++h) for (int t = 0; t < Tq; ++t) {
            const float *qr = q + (long)t * ldq + (long)h * hd;
            int kh = h / grp;
            int lim = causal ? (q_pos0 + t + 1) : Tk; if (lim > Tk) lim = Tk;
            float mx = -INFINITY;
            for (int j = 0; j < lim; ++j) {
                const float *kr = k + (long)j * ldkv + (long)kh * hd;
                float a = 0;
                #pragma omp simd reduction(+ : a)
                for (int i = 0; i < hd; ++i) a += qr[i] * kr[i];
                a *= scale; s[j] = a; if (a > mx) mx = a;
            }
            float l = 0; float acc[256]; for (int i = 0; i < hd; ++i) acc[i] = 0;
            for (int j = 0; j < lim; ++j) {
                float p = expf(s[j] - mx); l += p; p = RB(m, p);
                const float *vr = v + (long)j * ldkv + (long)kh * hd;
                for (int i = 0; i < hd; ++i) acc[i] += p * vr[i];
            }
            float *o = out + (long)t * ldo + (long)h * hd;
            for (int i = 0; i < hd; ++i) o[i] = RB(m, acc[i] / l);
        }
        free(s);
    }
}

/* ---------------------------------------------------------------- model */
enum { E_STEM = 0, E_PER_LAYER = 15, D_PER_LAYER = 9 };
static float **enc_layer_w(const oracle_model *m, int l) { return m->w + 4 + l * E_PER_LAYER; }
static float **enc_tail_w(const oracle_model *m) { return m->w + 4 + m->d.enc_layers * E_PER_LAYER; } /* norm.w, norm.b, proj l1.w,l1.b,l2.w,l2.b, embed */
static float **dec_layer_w(const oracle_model *m, int l) { return enc_tail_w(m) + 7 + l * D_PER_LAYER; }
static float *dec_final_norm(const oracle_model *m) { return *(enc_tail_w(m) + 7 + m->d.dec_layers * D_PER_LAYER); }

oracle_model *oracle_model_create(const oracle_dims *d, float **tensors, int n_tensors, int bf16) {
    int expect = 4 + d->enc_layers * E_PER_LAYER + 7 + d->dec_layers * D_PER_LAYER + 1;
    if (n_tensors != expect) { fprintf(stderr, "oracle: expected %d tensors, got %d\n", expect, n_tensors); return NULL; }
    oracle_model *m = (oracle_model *)calloc(1, sizeof(*m));
    m->d = *d; m->bf16 = bf16;
    m->w = (float **)malloc(sizeof(float *) * n_tensors);
    memcpy(m->w, tensors, sizeof(float *) * n_tensors);
    return m;
}
void oracle_model_destroy(oracle_model *m) { if (m) { free(m->w); free(m); } }

/* a7: conv stem.  feats [n_mels][n_frames] -> x [enc_T][enc_d] */
static void conv_stem(const oracle_model *m, const float *feats, float *x, oracle_outputs *o) {
    const oracle_dims *d = &m->d; int C = d->enc_d, F = d->n_frames, T = d->enc_T, M = d->n_mels;
    const float *w1 = m->w[0], *b1 = m->w[1], *w2 = m->w[2], *b2 = m->w[3];
    float *h1 = (float *)malloc(sizeof(float) * (long)C * F); /* [C][F] post-GELU */
    #pragma omp parallel for
    for (int c = 0; c < C; ++c) for (int t = 0; t < F; ++t) {
        float a = 0;
        for (int ci = 0; ci < M; ++ci) for (int k = 0; k < 3; ++k) {
            int tt = t + k - 1; if (tt < 0 || tt >= F) continue;
            a += RB(m, feats[(long)ci * F + tt]) * w1[((long)c * M + ci) * 3 + k];
        }
        a = RB(m, a + b1[c]);
        if (o && o->conv1) o->conv1[(long)c * F + t] = a;
        h1[(long)c * F + t] = RB(m, gelu_erf(a));
    }
    #pragma omp parallel for
    for (int c = 0; c < C; ++c) for (int t = 0; t < T; ++t) {
        float a = 0;
        for (int ci = 0; ci < C; ++ci) {
            const float *hr = h1 + (long)ci * F; const float *wr = w2 + ((long)c * C + ci) * 3;
            for (int k = 0; k < 3; ++k) { int tt = 2 * t + k - 1; if (tt < 0 || tt >= F) continue; a += hr[tt] * wr[k]; }
        }
        a = RB(m, a + b2[c]);
        if (o && o->conv2) o->conv2[(long)c * T + t] = a;
        x[(long)t * C + c] = RB(m, gelu_erf(a)); /* transpose(1,2) (:316) */
    }
    free(h1);
}

/* a8: one encoder layer in place on x [T][d] */
static void encoder_layer(const oracle_model *m, int l, float *x, int T) {
    const oracle_dims *d = &m->d; int D = d->enc_d, H = d->enc_heads, hd = D / H, FF = d->enc_ff;
    float **w = enc_layer_w(m, l);
    float *ln = (float *)malloc(sizeof(float) * (long)T * D), *q = (float *)malloc(sizeof(float) * (long)T * D);
    float *k = (float *)malloc(sizeof(float) * (long)T * D), *v = (float *)malloc(sizeof(float) * (long)T * D);
    float *a = (float *)malloc(sizeof(float) * (long)T * D), *ff = (float *)malloc(sizeof(float) * (long)T * FF);
    int *pos = (int *)malloc(sizeof(int) * T); for (int t = 0; t < T; ++t) pos[t] = t;
    layernorm(m, x, w[0], w[1], ln, T, D, d->enc_ln_eps);
    linear(m, ln, D, w[2], w[3], q, D, T, D, D);
    linear(m, ln, D, w[4], NULL, k, D, T, D, D);
    linear(m, ln, D, w[5], w[6], v, D, T, D, D);
    rope(m, q, T, H, hd, d->enc_rotary_dim, d->enc_theta, pos);
    rope(m, k, T, H, hd, d->enc_rotary_dim, d->enc_theta, pos);
    attention(m, q, D, k, v, D, a, D, T, T, H, H, hd, 0, 0);
    linear(m, a, D, w[7], w[8], q, D, T, D, D);
    for (long i = 0; i < (long)T * D; ++i) x[i] = RB(m, x[i] + q[i]);
    layernorm(m, x, w[9], w[10], ln, T, D, d->enc_ln_eps);
    linear(m, ln, D, w[11], w[12], ff, FF, T, FF, D);
    for (long i = 0; i < (long)T * FF; ++i) ff[i] = RB(m, gelu_erf(ff[i]));
    linear(m, ff, FF, w[13], w[14], q, D, T, D, FF);
    for (long i = 0; i < (long)T * D; ++i) x[i] = RB(m, x[i] + q[i]);
    free(ln); free(q); free(k); free(v); free(a); free(ff); free(pos);
}

static int floordiv_i(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
/* a7-a9: feats -> audio embeds [n_keep][dec_d]; returns rows kept */
int oracle_audio_features(const oracle_model *m, const float *feats, int n_valid_frames, float *embeds, oracle_outputs *o) {
    const oracle_dims *d = &m->d; int T = d->enc_T, D = d->enc_d;
    float *x = (float *)malloc(sizeof(float) * (long)T * D);
    conv_stem(m, feats, x, o);
    for (int l = 0; l < d->enc_layers; ++l) {
        encoder_layer(m, l, x, T);
        if (o && o->enc_layers) memcpy(o->enc_layers + (long)l * T * D, x, sizeof(float) * (long)T * D);
    }
    float **tw = enc_tail_w(m);
    float *y = (float *)malloc(sizeof(float) * (long)T * D);
    layernorm(m, x, tw[0], tw[1], y, T, D, d->enc_ln_eps);
    if (o && o->enc_out) memcpy(o->enc_out, y, sizeof(float) * (long)T * D);
    int Tm = T / d->merge, PI = D * d->merge, PM = d->dec_d * 2;
    float *h = (float *)malloc(sizeof(float) * (long)Tm * PM);
    linear(m, y, PI, tw[2], tw[3], h, PM, Tm, PM, PI); /* reshape [T][D] -> [T/4][4D] is a view */
    for (long i = 0; i < (long)Tm * PM; ++i) h[i] = RB(m, gelu_erf(h[i]));
    float *e = (float *)malloc(sizeof(float) * (long)Tm * d->dec_d);
    linear(m, h, PM, tw[4], tw[5], e, d->dec_d, Tm, d->dec_d, PM);
    int L = n_valid_frames; /* modeling_glmasr.py:399-403, python floor division */
    L = floordiv_i(L + 2 - 2 - 1, 1) + 1; L = floordiv_i(L + 2 - 2 - 1, 2) + 1;
    int keep = floordiv_i(L - d->merge, d->merge) + 1; if (keep < 0) keep = 0; if (keep > Tm) keep = Tm;
    memcpy(embeds, e, sizeof(float) * (long)keep * d->dec_d);
    free(x); free(y); free(h); free(e);
    return keep;
}

typedef struct { float *k, *v; int cap; } kv_cache; /* per layer [cap][kv_dim] */

/* one decoder pass over n new tokens whose embeddings are in x [n][dec_d]; cache holds `past` tokens */
static void decoder_forward(const oracle_model *m, float *x, int n, int past, kv_cache *kv, float *layer_out /* [L][n][d] or NULL */) {
    const oracle_dims *d = &m->d; int D = d->dec_d, Hq = d->dec_heads, Hkv = d->dec_kv_heads, hd = d->dec_head_dim, FF = d->dec_ff;
    int QD = Hq * hd, KD = Hkv * hd;
    float *hn = (float *)malloc(sizeof(float) * (long)n * D), *q = (float *)malloc(sizeof(float) * (long)n * QD);
    float *a = (float *)malloc(sizeof(float) * (long)n * QD), *o = (float *)malloc(sizeof(float) * (long)n * D);
    float *g = (float *)malloc(sizeof(float) * (long)n * FF), *u = (float *)malloc(sizeof(float) * (long)n * FF);
    int *pos = (int *)malloc(sizeof(int) * n); for (int t = 0; t < n; ++t) pos[t] = past + t;
    for (int l = 0; l < d->dec_layers; ++l) {
        float **w = dec_layer_w(m, l);
        rmsnorm(m, x, w[0], hn, n, D, d->dec_rms_eps);
        linear(m, hn, D, w[1], NULL, q, QD, n, QD, D);
        float *kn = kv[l].k + (long)past * KD, *vn = kv[l].v + (long)past * KD;
        linear(m, hn, D, w[2], NULL, kn, KD, n, KD, D);
        linear(m, hn, D, w[3], NULL, vn, KD, n, KD, D);
        rope(m, q, n, Hq, hd, hd, d->dec_theta, pos);
        rope(m, kn, n, Hkv, hd, hd, d->dec_theta, pos);
        /* sdpa_attention_forward: is_causal only when q_len > 1 (sdpa_attention.py) */
        attention(m, q, QD, kv[l].k, kv[l].v, KD, a, QD, n, past + n, Hq, Hkv, hd, n > 1, past);
        linear(m, a, QD, w[4], NULL, o, D, n, D, QD);
        for (long i = 0; i < (long)n * D; ++i) x[i] = RB(m, x[i] + o[i]);
        rmsnorm(m, x, w[5], hn, n, D, d->dec_rms_eps);
        linear(m, hn, D, w[6], NULL, g, FF, n, FF, D);
        linear(m, hn, D, w[7], NULL, u, FF, n, FF, D);
        for (long i = 0; i < (long)n * FF; ++i) g[i] = RB(m, RB(m, silu(g[i])) * u[i]);
        linear(m, g, FF, w[8], NULL, o, D, n, D, FF);
        for (long i = 0; i < (long)n * D; ++i) x[i] = RB(m, x[i] + o[i]);
        if (layer_out) memcpy(layer_out + (long)l * n * D, x, sizeof(float) * (long)n * D);
    }
    free(hn); free(q); free(a); free(o); free(g); free(u); free(pos);
}

static void lm_head(const oracle_model *m, const float *x /* [dec_d] */, float *logits) {
    const oracle_dims *d = &m->d;
    float *hn = (float *)malloc(sizeof(float) * d->dec_d);
    rmsnorm(m, x, dec_final_norm(m), hn, 1, d->dec_d, d->dec_rms_eps);
    linear(m, hn, d->dec_d, enc_tail_w(m)[6], NULL, logits, d->vocab, 1, d->vocab, d->dec_d); /* tied (modeling_glmasr.py:517) */
    free(hn);
}
static int argmax_first(const float *x, int n) { int b = 0; for (int i = 1; i < n; ++i) if (x[i] > x[b]) b = i; return b; }

/* Full path for one request of W >= 1 windows (HF:processing_glmasr.py:136-157 cuts audio longer than 30 s into windows; the
 * encoder runs per window and the kept rows of all windows are concatenated, modeling_glmasr.py:380-408).  feats: [W][n_mels][n_frames].
 * Per-stage taps (conv*, enc_layers, enc_out) record the LAST window.  Returns 0, or -1 when placeholders != audio rows (:426-429). */
int oracle_transcribe_multi(const oracle_model *m, const float *feats, const int *n_valid_frames, int W, const int *prompt, int P,
                            int max_new, oracle_outputs *o) {
    const oracle_dims *d = &m->d; int D = d->dec_d, KD = d->dec_kv_heads * d->dec_head_dim;
    int Tm = d->enc_T / d->merge;
    float *emb = (float *)malloc(sizeof(float) * (long)W * Tm * D);
    int n_audio = 0;
    for (int w = 0; w < W; ++w)
        n_audio += oracle_audio_features(m, feats + (long)w * d->n_mels * d->n_frames, n_valid_frames[w], emb + (long)n_audio * D, o);
    if (o && o->audio_embeds) memcpy(o->audio_embeds, emb, sizeof(float) * (long)n_audio * D);
    int n_ph = 0; for (int i = 0; i < P; ++i) n_ph += (prompt[i] == d->audio_token_id);
    if (n_ph != n_audio) { free(emb); return -1; }
    const float *E = enc_tail_w(m)[6];
    float *x = (float *)malloc(sizeof(float) * (long)P * D);
    for (int i = 0, a = 0; i < P; ++i) {
        if (prompt[i] == d->audio_token_id) memcpy(x + (long)i * D, emb + (long)(a++) * D, sizeof(float) * D);
        else memcpy(x + (long)i * D, E + (long)prompt[i] * D, sizeof(float) * D);
    }
    int cap = P + max_new + 1;
    kv_cache *kv = (kv_cache *)malloc(sizeof(kv_cache) * d->dec_layers);
    for (int l = 0; l < d->dec_layers; ++l) { kv[l].k = (float *)calloc((long)cap * KD, 4); kv[l].v = (float *)calloc((long)cap * KD, 4); kv[l].cap = cap; }
    decoder_forward(m, x, P, 0, kv, o ? o->dec_layers : NULL);
    float *logits = (float *)malloc(sizeof(float) * d->vocab);
    lm_head(m, x + (long)(P - 1) * D, logits);
    if (o && o->prefill_logits) memcpy(o->prefill_logits, logits, sizeof(float) * d->vocab);
    int n_new = 0, past = P; float *xt = (float *)malloc(sizeof(float) * D);
    for (int step = 0; step < max_new; ++step) { /* generation/utils.py:2876-2943 */
        if (o && o->step_logits) memcpy(o->step_logits + (long)step * d->vocab, logits, sizeof(float) * d->vocab);
        int tok = argmax_first(logits, d->vocab);
        if (o && o->force_ids) tok = o->force_ids[step];
        if (o && o->new_ids) o->new_ids[step] = tok;
        ++n_new;
        int stop = 0; for (int e = 0; e < d->n_eos; ++e) stop |= (tok == d->eos[e]);
        if (stop || step == max_new - 1) break;
        memcpy(xt, E + (long)tok * D, sizeof(float) * D);
        decoder_forward(m, xt, 1, past, kv, NULL); ++past;
        lm_head(m, xt, logits);
    }
    if (o && o->n_new) *o->n_new = n_new;
    for (int l = 0; l < d->dec_layers; ++l) { free(kv[l].k); free(kv[l].v); }
    free(kv); free(emb); free(x); free(logits); free(xt);
    return 0;
}
int oracle_transcribe(const oracle_model *m, const float *feats, int n_valid_frames, const int *prompt, int P,
                      int max_new, oracle_outputs *o) {
    return oracle_transcribe_multi(m, feats, &n_valid_frames, 1, prompt, P, max_new, o);
}

/* ---------------------------------------------------------------- single-op entry points (full-size stage checks) */
void oracle_linear(const float *x, const float *w, const float *b, float *y, int T, int N, int K, int bf16) {
    oracle_model m; memset(&m, 0, sizeof(m)); m.bf16 = bf16; linear(&m, x, K, w, b, y, N, T, N, K);
}
void oracle_gelu(float *x, long n, int bf16) { for (long i = 0; i < n; ++i) { float v = gelu_erf(x[i]); x[i] = bf16 ? bf16_round(v) : v; } }
void oracle_attention(const float *q, const float *k, const float *v, float *out, int Tq, int Tk, int Hq, int Hkv, int hd,
                      int causal, int q_pos0, int bf16) {
    oracle_model m; memset(&m, 0, sizeof(m)); m.bf16 = bf16;
    attention(&m, q, (long)Hq * hd, k, v, (long)Hkv * hd, out, (long)Hq * hd, Tq, Tk, Hq, Hkv, hd, causal, q_pos0);
}
void oracle_rope(float *x, int T, int heads, int hd, int rd, float theta, const int *pos, int bf16) {
    oracle_model m; memset(&m, 0, sizeof(m)); m.bf16 = bf16; rope(&m, x, T, heads, hd, rd, theta, pos);
}
void oracle_layernorm(const float *x, const float *w, const float *b, float *y, int T, int d, float eps, int bf16) {
    oracle_model m; memset(&m, 0, sizeof(m)); m.bf16 = bf16; layernorm(&m, x, w, b, y, T, d, eps);
}
void oracle_rmsnorm(const float *x, const float *w, float *y, int T, int d, float eps, int bf16) {
    oracle_model m; memset(&m, 0, sizeof(m)); m.bf16 = bf16; rmsnorm(&m, x, w, y, T, d, eps);
}
void oracle_encoder_layer(const oracle_model *m, int l, float *x, int T) { encoder_layer(m, l, x, T); }
