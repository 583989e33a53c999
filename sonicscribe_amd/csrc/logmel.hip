// Log-mel front-end: int16 PCM -> Whisper log-mel features.
// Follows HF:models/whisper/feature_extraction_whisper.py:135-168,300-341 (zero-pad to 30 s, reflect-padded
// centred STFT n_fft=400 hop=160 periodic hann, |.|^2, drop last frame, slaney mel bank, log10 clamp,
// per-segment max-8 floor, (x+4)/4) in fp32.
//
// logmel_power_kernel: one block = 32 consecutive frames of one segment.
//   * PCM window (31*160+400 = 5360 samples) loaded once with coalesced int16 loads, converted to fp32 in LDS
//     (frame stride skewed to 161 words so the 32 frame rows of an MFMA A-operand read hit 32 banks);
//   * windowed real DFT as an exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32) on inputs folded by both symmetries of the real
//     400-point transform (see the kernel): four 100-long sums for bins 0..100, bins 101..200 by mirroring;
//     the twiddle matrix is never materialised: B-operand values come from 400-entry cos/sin tables in LDS,
//     indexed (n*bin) mod 400 with an incremental index;
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


// in-kernel timeline for tools/mel_timeline.hip (compiled out of the product library)
#ifdef LM_TRACE
__device__ long long* lm_trace;
#define LMT(k) do { if (threadIdx.x == 0) lm_trace[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define LMT(k) do { } while (0)
#endif

__device__ __forceinline__ int ord_enc(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord_dec(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

// DFT with both symmetries of a real 400-point transform folded out before the GEMM (4x fewer MACs than the plain
// frames[32 x 400] . twiddle[400 x 402] product).  With y = window * samples:
//   E[n] = y[n] + y[400-n],  O[n] = y[n] - y[400-n]   (n = 1..199)            input symmetry of cos / sin
//   Re X[k] = y[0] + (-1)^k y[200] + sum_n E[n] cos(2 pi k n / 400),   Im X[k] = - sum_n O[n] sin(2 pi k n / 400)
//   cos(2 pi (200-k) n / 400) = (-1)^n cos(2 pi k n / 400),  sin(...) = -(-1)^n sin(...)          bin symmetry k <-> 200-k
// so with the sums split by the parity of n,  Ae/Ao = sum over even/odd n of E cos,  Be/Bo = ... of O sin  (k = 0..100):
//   Re X[k] = c + Ae + Ao,  Re X[200-k] = c + Ae - Ao,  |Im X[k]| = |Be + Bo|,  |Im X[200-k]| = |Be - Bo|,  c = y[0] + (-1)^k y[200].
// Four 100-long sums for 101 bins: wave w owns bins 32w .. 32w+31, four fp32 MFMA accumulators, 50 k-steps of v_mfma_f32_32x32x2_f32.
#define LM_FOLD_LD 401                 // per-frame stride of the folded vectors [4][100] (+1: 32 frame rows hit 32 banks)
#define LM_DYN_LDS ((LM_SMP + 40 + LM_FT * LM_FOLD_LD + 3 * LM_NFFT + 2 * LM_FT) * 4)

__global__ __launch_bounds__(256) void logmel_power_kernel(const int16_t* pcm, long pcm_stride, const int* n_samples, LogmelConst lc,
                                                            float* logspec /* [B][n_frames][n_mels] */, int* segmax, int n_frames, int n_mels) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* smp = lds;                                   // [LM_SMP + 40] raw samples, frame stride skewed to 161
    float* fold = smp + LM_SMP + 40;                    // [32][4][100]: E even-n, E odd-n, O even-n, O odd-n
    float* s_win = fold + LM_FT * LM_FOLD_LD;
    float* s_cos = s_win + LM_NFFT;
    float* s_sin = s_cos + LM_NFFT;
    float* s_y0 = s_sin + LM_NFFT;                      // [32] y[0], then [32] y[200]
    float* pw = lds;                                    // [32][LM_PWLD] power spectrum, aliases smp/fold after the DFT
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.y, t0 = blockIdx.x * LM_FT;
    const int n_pad = n_frames * LM_HOP;
    int n = n_samples[b]; n = n < n_pad ? n : n_pad;
    // frames t >= t_live see only zero padding:  t*160 - 200 >= n
    const int t_live = min(n_frames, (n + LM_NFFT / 2 + LM_HOP - 1) / LM_HOP);
    if (t0 >= t_live) return;
    const int16_t* x = pcm + (long)b * pcm_stride;
    LMT(0);

    // PCM window -> LDS as fp32.  16-byte loads (8 samples per lane; the window starts at a multiple of 8 samples and the segment base is 16-byte
    // aligned) wherever the 8 samples lie inside [0, n): 3 trips instead of the 21 dependent 2-byte trips of the first form; the reflected edges
    // (torch.stft center=True) and the zero padding behind the audio keep the per-sample path.
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int base = t0 * LM_HOP - LM_NFFT / 2;
    const bool vec_ok = (pcm_stride & 7) == 0 && ((size_t)pcm & 15) == 0;
    for (int v8 = tid; v8 < LM_SMP / 8; v8 += 256) {
        const int i0 = v8 * 8, j0 = base + i0;
        if (vec_ok && j0 >= 0 && j0 + 8 <= n) {
            const s16x8 q = *(const s16x8*)(x + j0);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + k; smp[i + i / LM_HOP] = (float)q[k] * (1.0f / 32768.0f); }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k;
                int j = base + i;
                if (j < 0) j = -j;                              // reflect (torch.stft center=True)
                if (j >= n_pad) j = 2 * (n_pad - 1) - j;
                smp[i + i / LM_HOP] = (j < n) ? (float)x[j] * (1.0f / 32768.0f) : 0.0f;   // skew: frame stride 161
            }
        }
    }
    for (int i = tid; i < LM_NFFT; i += 256) { s_win[i] = lc.win[i]; s_cos[i] = lc.cos_t[i]; s_sin[i] = lc.sin_t[i]; }
    __syncthreads();
    LMT(1);
    // fold: one (frame, n) pair per item, n = 1..199; n = 0 and n = 200 go to s_y0
#pragma unroll 5
    for (int i = tid; i < LM_FT * 200; i += 256) {          // 25 trips; unrolled so that five trips' LDS reads are in flight together
        const int f = i / 200, nn = i % 200;
        const float* fs = smp + f * (LM_HOP + 1);
        if (nn == 0) {
            s_y0[f] = fs[0] * s_win[0];
            s_y0[LM_FT + f] = fs[200 + 200 / LM_HOP] * s_win[200];
            fold[f * LM_FOLD_LD + 0] = 0.f; fold[f * LM_FOLD_LD + 200] = 0.f;       // the j = 0 slots of the even-n vectors
        } else {
            const int m = LM_NFFT - nn;
            const float ya = fs[nn + nn / LM_HOP] * s_win[nn], yb = fs[m + m / LM_HOP] * s_win[m];
            const int cls = nn & 1, j = nn >> 1;
            fold[f * LM_FOLD_LD + cls * 100 + j] = ya + yb;                          // E
            fold[f * LM_FOLD_LD + 200 + cls * 100 + j] = ya - yb;                    // O
        }
    }
    __syncthreads();
    LMT(2);

    const int fi = lane & 31, kh = lane >> 5;
    const int bin = wid * 32 + fi;
    f32x16 ae, ao, be, bo;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ae[r] = 0.f; ao[r] = 0.f; be[r] = 0.f; bo[r] = 0.f; }
    int ie = (bin * (2 * kh)) % LM_NFFT, io = (bin * (2 * kh + 1)) % LM_NFFT;       // table index of n = 4s + 2kh (+1)
    const int stp = (4 * bin) % LM_NFFT;
    const float* fr = fold + fi * LM_FOLD_LD;
    for (int s2 = 0; s2 < 50; ++s2) {
        const int j = 2 * s2 + kh;
        const float e0 = fr[j], e1 = fr[100 + j], o0 = fr[200 + j], o1 = fr[300 + j];
        const float ce = s_cos[ie], se = s_sin[ie], co = s_cos[io], so = s_sin[io];
        ae = __builtin_amdgcn_mfma_f32_32x32x2f32(e0, ce, ae, 0, 0, 0);
        ao = __builtin_amdgcn_mfma_f32_32x32x2f32(e1, co, ao, 0, 0, 0);
        be = __builtin_amdgcn_mfma_f32_32x32x2f32(o0, se, be, 0, 0, 0);
        bo = __builtin_amdgcn_mfma_f32_32x32x2f32(o1, so, bo, 0, 0, 0);
        ie += stp; ie = ie >= LM_NFFT ? ie - LM_NFFT : ie;
        io += stp; io = io >= LM_NFFT ? io - LM_NFFT : io;
    }
    LMT(3);
    // y[0], y[200] of this lane's 16 frames before pw overwrites the staging area
    float cc[16];
    const float sgn = (bin & 1) ? -1.0f : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * kh;
        cc[r] = s_y0[f] + sgn * s_y0[LM_FT + f];
    }
    __syncthreads();                                    // every wave is done with fold / smp: pw may overwrite them
    // C/D map of the 32x32 MFMA: col = lane & 31 (bin), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (frame)
    if (bin <= 100) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float re1 = cc[r] + ae[r] + ao[r], im1 = be[r] + bo[r];
            const float m1 = sqrtf(re1 * re1 + im1 * im1);                          // stft.abs() ** 2
            pw[f * LM_PWLD + bin] = m1 * m1;
            if (bin < 100) {
                const float re2 = cc[r] + ae[r] - ao[r], im2 = be[r] - bo[r];
                const float m2 = sqrtf(re2 * re2 + im2 * im2);
                pw[f * LM_PWLD + 200 - bin] = m2 * m2;
            }
        }
    }
    __syncthreads();
    LMT(4);

    // ---- mel + log10: thread = (mel, frame half).  The filter's taps (<= ~27 of them for the 128-filter slaney bank) are fetched ONCE into registers
    // and reused for the thread's 16 frames: the first form re-read them from global memory for every frame (432 dependent-ish loads per thread, the
    // longest phase of the kernel); same summation order, same bits.
    float lmax = -1e30f;
    constexpr int MT = 32;
    for (int e = tid; e < n_mels * 2; e += 256) {
        const int m = e % n_mels, fh = e / n_mels;
        const int lo = lc.mel_lo[m], cnt = lc.mel_cnt[m];
        const float* w = lc.mel_w + lc.mel_off[m];
        float wr[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) wr[i] = i < cnt ? w[i] : 0.f;
        if (cnt <= MT) {
            // taps outer, the thread's 16 frames inner: 16 independent accumulators per tap instead of one 27-long dependent FMA chain per frame
            // (each output still sums its taps in ascending order: same bits)
            float acc[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[f] = 0.f;
            const float* pr = pw + fh * 16 * LM_PWLD + lo;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < cnt) {
#pragma unroll
                    for (int f = 0; f < 16; ++f) acc[f] += wr[i] * pr[f * LM_PWLD + i];
                }
            }
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                const int t = t0 + fh * 16 + f;
                if (t < t_live) {
                    const float v = log10f(fmaxf(acc[f], 1e-10f));
                    logspec[((long)b * n_frames + t) * n_mels + m] = v;
                    lmax = fmaxf(lmax, v);
                }
            }
        } else {
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
    }
    lmax = wave_max(lmax);
    if (lane == 0 && lmax > -1e29f) atomicMax(&segmax[b], ord_enc(lmax));
    LMT(5);
}

template <typename T>
__global__ void logmel_finalize_kernel(const float* logspec, const int* segmax, const int* n_samples, int n_frames, int n_mels,
                                       T* feats_fm /* [B][n_frames+2][n_mels], rows 0 and n_frames+1 stay zero */,
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
    feats_fm[((long)b * (n_frames + 2) + 1 + t) * n_mels + m] = (T)v;      // asr.py:280-301: features cast to the model dtype
    if (feats_f32) feats_f32[((long)b * n_mels + m) * n_frames + t] = v;
}

void launch_logmel(const int16_t* pcm, long pcm_stride, const int* n_samples_dev, int max_samples, const LogmelConst& lc,
                   float* logspec, int* segmax, int B, int n_frames, int n_mels, bf16_t* feats_fm, float* feats_f32, hipStream_t s, int dt) {
    launch_fill_i32(segmax, (int)0x80808080, B, s);   // ordered-int encoding of a very negative float
    const int n_pad = n_frames * LM_HOP;
    const int n = max_samples < n_pad ? max_samples : n_pad;
    int t_live = (n + LM_NFFT / 2 + LM_HOP - 1) / LM_HOP; if (t_live > n_frames) t_live = n_frames;
    const int tiles = (t_live + LM_FT - 1) / LM_FT;
    ensure_dyn_lds((const void*)logmel_power_kernel, LM_DYN_LDS);
    if (tiles > 0) hipLaunchKernelGGL(logmel_power_kernel, dim3(tiles, B), dim3(256), LM_DYN_LDS, s, pcm, pcm_stride, n_samples_dev, lc, logspec, segmax, n_frames, n_mels);
    const long tot = (long)n_frames * n_mels;
    DT_SWITCH(dt, T, hipLaunchKernelGGL(logmel_finalize_kernel<T>, dim3((tot + 255) / 256, B), dim3(256), 0, s, logspec, segmax, n_samples_dev, n_frames, n_mels, (T*)feats_fm, feats_f32));
}
