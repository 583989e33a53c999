// Log-mel front-end: int16 PCM -> Whisper log-mel features.
// Follows HF:models/whisper/feature_extraction_whisper.py:135-168,300-341 (zero-pad to 30 s, reflect-padded
// centred STFT n_fft=400 hop=160 periodic hann, |.|^2, drop last frame, slaney mel bank, log10 clamp,
// per-segment max-8 floor, (x+4)/4) in fp32.
//
// logmel_power_kernel: one block = 32 consecutive frames of one segment.
//   * PCM window (31*160+400 = 5360 samples) loaded once with coalesced int16 loads, converted to fp32 in LDS
//     (frame stride skewed to 161 words so the 32 frame rows of an MFMA A-operand read hit 32 banks);
//   * windowed real DFT as an exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32): frames[32 x 400] . twiddle[400 x bins];
//     the twiddle matrix is never materialised: B-operand values come from 400-entry cos/sin tables in LDS,
//     indexed (k*bin) mod 400 with an incremental index;
//   * power -> LDS, sparse triangular mel bank (CSR, <= ~27 taps per filter), log10, per-segment max via
//     one ordered-int atomicMax per wave.
//   Frames that lie entirely in the zero padding are skipped (their log-mel is the clamp floor).
// logmel_finalize_kernel: applies the max-8 floor and (x+4)/4, emits bf16 frame-major features with one zero
// row of padding on each side (the conv stem consumes them as an im2col-free GEMM operand) and, for the parity
// API, the fp32 [128][3000] HF layout.
#include "common.h"
#include "kernels.h"

#define LM_NFFT 400
#define LM_HOP 160
#define LM_BINS 201
#define LM_FT 32                       // frames per block
#define LM_SMP (31 * LM_HOP + LM_NFFT) // 5360 samples per block
#define LM_PWLD 225                    // power row stride (7 bin blocks of 32 = 224, +1 skew)


__device__ __forceinline__ int ord_enc(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord_dec(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(256) void logmel_power_kernel(const int16_t* pcm, long pcm_stride, const int* n_samples, LogmelConst lc,
                                                            float* logspec /* [B][n_frames][n_mels] */, int* segmax, int n_frames, int n_mels) {
    __shared__ float smp[LM_SMP + 40];
    __shared__ float s_win[LM_NFFT], s_cos[LM_NFFT], s_sin[LM_NFFT];
    __shared__ float pw[LM_FT * LM_PWLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * LM_FT;
    const int n_pad = n_frames * LM_HOP;
    int n = n_samples[b]; n = n < n_pad ? n : n_pad;
    // frames t >= t_live see only zero padding:  t*160 - 200 >= n
    const int t_live = min(n_frames, (n + LM_NFFT / 2 + LM_HOP - 1) / LM_HOP);
    if (t0 >= t_live) return;
    const int16_t* x = pcm + (long)b * pcm_stride;

    for (int i = tid; i < LM_SMP; i += 256) {
        int j = t0 * LM_HOP - LM_NFFT / 2 + i;
        if (j < 0) j = -j;                                  // reflect (torch.stft center=True)
        if (j >= n_pad) j = 2 * (n_pad - 1) - j;
        const float v = (j < n) ? (float)x[j] * (1.0f / 32768.0f) : 0.0f;
        smp[i + i / LM_HOP] = v;                            // skew: frame stride 161
    }
    for (int i = tid; i < LM_NFFT; i += 256) { s_win[i] = lc.win[i]; s_cos[i] = lc.cos_t[i]; s_sin[i] = lc.sin_t[i]; }
    __syncthreads();

    // ---- DFT: wave w owns bin blocks {w, w+4} (7 blocks of 32 bins cover 0..223 >= 201)
    const int fi = lane & 31, kh = lane >> 5;
    f32x16 re[2], im[2];
    int idx[2], stp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { re[q][r] = 0.f; im[q][r] = 0.f; }
        const int bin = (wid + 4 * q) * 32 + fi;
        idx[q] = (kh * bin) % LM_NFFT;
        stp[q] = (2 * bin) % LM_NFFT;
    }
    const int nblk = (wid + 4 < 7) ? 2 : 1;
    const int abase = fi * (LM_HOP + 1);
    for (int s = 0; s < LM_NFFT / 2; ++s) {
        const int k = 2 * s + kh;
        const float av = smp[abase + k + k / LM_HOP] * s_win[k];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q < nblk) {
                const float cv = s_cos[idx[q]], sv = s_sin[idx[q]];
                re[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, cv, re[q], 0, 0, 0);
                im[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sv, im[q], 0, 0, 0);
                idx[q] += stp[q];
                idx[q] = idx[q] >= LM_NFFT ? idx[q] - LM_NFFT : idx[q];
            }
        }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31 (bin), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (frame)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q < nblk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int fr = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float mag = sqrtf(re[q][r] * re[q][r] + im[q][r] * im[q][r]);   // stft.abs() ** 2
                pw[fr * LM_PWLD + (wid + 4 * q) * 32 + fi] = mag * mag;
            }
        }
    }
    __syncthreads();

    // ---- mel + log10: thread = (mel, frame half)
    float lmax = -1e30f;
    for (int e = tid; e < n_mels * 2; e += 256) {
        const int m = e % n_mels, fh = e / n_mels;
        const int lo = lc.mel_lo[m], cnt = lc.mel_cnt[m];
        const float* w = lc.mel_w + lc.mel_off[m];
        for (int f = fh * 16; f < fh * 16 + 16; ++f) {
            const int t = t0 + f;
            if (t >= t_live) break;
            float acc = 0.f;
            for (int i = 0; i < cnt; ++i) acc += w[i] * pw[f * LM_PWLD + lo + i];
            const float v = log10f(fmaxf(acc, 1e-10f));
            logspec[((long)b * n_frames + t) * n_mels + m] = v;
            lmax = fmaxf(lmax, v);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0 && lmax > -1e29f) atomicMax(&segmax[b], ord_enc(lmax));
}

__global__ void logmel_finalize_kernel(const float* logspec, const int* segmax, const int* n_samples, int n_frames, int n_mels,
                                       bf16_t* feats_fm /* [B][n_frames+2][n_mels], rows 0 and n_frames+1 stay zero */,
                                       float* feats_f32 /* optional [B][n_mels][n_frames] */) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n_frames * n_mels) return;
    const int t = e / n_mels, m = e % n_mels;
    const int n_pad = n_frames * LM_HOP;
    int n = n_samples[b]; n = n < n_pad ? n : n_pad;
    const int t_live = min(n_frames, (n + LM_NFFT / 2 + LM_HOP - 1) / LM_HOP);
    const float floor_v = log10f(1e-10f);
    float gmax = ord_dec(segmax[b]);
    if (t_live < n_frames || gmax < floor_v) gmax = fmaxf(gmax, floor_v);
    float v = t < t_live ? logspec[((long)b * n_frames + t) * n_mels + m] : floor_v;
    v = fmaxf(v, gmax - 8.0f);
    v = (v + 4.0f) / 4.0f;
    feats_fm[((long)b * (n_frames + 2) + 1 + t) * n_mels + m] = f2bf(v);
    if (feats_f32) feats_f32[((long)b * n_mels + m) * n_frames + t] = v;
}

void launch_logmel(const int16_t* pcm, long pcm_stride, const int* n_samples_dev, int max_samples, const LogmelConst& lc,
                   float* logspec, int* segmax, int B, int n_frames, int n_mels, bf16_t* feats_fm, float* feats_f32, hipStream_t s) {
    launch_fill_i32(segmax, (int)0x80808080, B, s);   // ordered-int encoding of a very negative float
    const int n_pad = n_frames * LM_HOP;
    const int n = max_samples < n_pad ? max_samples : n_pad;
    int t_live = (n + LM_NFFT / 2 + LM_HOP - 1) / LM_HOP; if (t_live > n_frames) t_live = n_frames;
    const int tiles = (t_live + LM_FT - 1) / LM_FT;
    if (tiles > 0) hipLaunchKernelGGL(logmel_power_kernel, dim3(tiles, B), dim3(256), 0, s, pcm, pcm_stride, n_samples_dev, lc, logspec, segmax, n_frames, n_mels);
    const long tot = (long)n_frames * n_mels;
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3((tot + 255) / 256, B), dim3(256), 0, s, logspec, segmax, n_samples_dev, n_frames, n_mels, feats_fm, feats_f32);
}
