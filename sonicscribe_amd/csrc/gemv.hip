// gemv.hip: the token step at 1 - 4 rows as a chain of weight-streaming GEMV kernels (round 6; option decode_gemv).
// MEASURED AND LOST (profiles/round6_gemv_ab.txt): correct, deterministic, within one ulp of the MFMA chain - and 6 % slower per layer at one row (38.6 against 36.3 us;
// B = 1 call 198.8 against 179.3 ms per 20 s / 150 tokens), because at one row the MFMA kernels are already bound by the weight stream, not by their tile machinery
// (gate/up: 50 MB in 10.7 us = 4.7 TB/s; the same per-layer time as the guide's five-kernel "launches-baseline" when scaled to this layer's bytes).  Kept selectable and tested,
// like the other experiments that lost; the expectation it was built on is left below as written.
//
// Why: BASELINE configs 1 and 5 are latency shapes - one `transcribe()` call at a time (backend/transcription_manager.py:43-65), a 20 s final of one session.  At one row the
// MFMA decode kernels of gemm.hip still stage a 16-row X image, reduce 16 x 16 accumulator tiles through LDS and hand K slabs to a separate add + RMSNorm launch: 1.19 ms per token
// step, 42 us per layer, barely less than at 32 rows.  A row count this small needs none of that: the weights go through every lane once (v_dot2c on the 16-byte fragments of the
// SAME fragment-tiled copies the MFMA kernels read - one weight copy), the few activation rows sit in LDS, every projection sees its whole K inside one block (no slabs), and the
// norms ride in the consumer.  Five launches per layer:
//     q|k|v  = (RMSNorm(x) . w_ln1) Wqkv^T                       -> fp32 [rows][3072], consumed by decode_attn_kernel as ONE "slab" (RoPE, KV append, attention unchanged)
//     x     += attention_out Wo^T                                  (residual add in the epilogue, modeling_llama.py:306-309)
//     act    = silu(g) * u,  (g, u) = (RMSNorm(x) . w_ln2) Wgu^T   (modeling_llama.py:163-176; 8-row gate/up interleave of the tiled copy)
//     x     += act Wdown^T
//   and behind the last layer  logits = (RMSNorm(x) . w_norm) E^T  (tied lm_head) -> greedy_kernel.
// Rounding points are the reference's (bf16 / fp16 after every module output, RMSNorm's two roundings: modeling_llama.py:60-65); what differs from the MFMA chain is the order in which
// an output's K products are added (per lane over its k-steps, four lanes, eight waves - fixed, so results are deterministic), i.e. isolated last-bit flips of a bf16 logit: the
// same class of difference as between the reference's own batch sizes.  That is why this chain is an OPTION only: with it a request's low bits depend on whether its step ran with <= 4 rows,
// which the default configuration guarantees never to be the case (batch invariance).
//
// Block = TPB weight tiles of 16 output rows, the whole K; wave w owns K eighth w (KS8 k-steps of 32): its W fragments are requested up front (nontemporal), its slice of the
// activation rows is staged (and normalised) by itself, so nothing waits for another wave until the one reduction over [8 waves][TPB][rows][16].
#include "common.h"
#include "kernels.h"

template <typename T> struct Dot2;
template <> struct Dot2<bf16_t> {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    static __device__ __forceinline__ float f(v2 a, v2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
};
template <> struct Dot2<f16_t> {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    static __device__ __forceinline__ float f(v2 a, v2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
};

template <typename T, int MODE, int TPB, int KS8>
__global__ __launch_bounds__(512) void gemv_kernel(GemvArgs a) {
    typedef typename ET<T>::v8 V8;
    typedef typename Dot2<T>::v2 V2;
    constexpr int K = KS8 * 256, KE = K / 8, RMAX = GEMV_MAX_ROWS;       // KE: elements of a wave's K eighth
    constexpr bool NORM = MODE != GEMV_RESID;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = (T*)smem;                                                     // [R][K] activation rows (NORM: normalised)
    float* red = (float*)(smem + (size_t)RMAX * K * sizeof(T));           // [8][TPB][RMAX][16]
    __shared__ float part[RMAX][8];
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6, R = a.M;
    const T* W = (const T*)a.W + ((long)blockIdx.x * TPB * (K >> 5) + wk * KS8) * 512 + lane * 8;
    // weights first: the whole K eighth of every tile of the block in flight at once
    V8 wf[TPB][KS8];
#pragma unroll
    for (int j = 0; j < TPB; ++j)
#pragma unroll
        for (int u = 0; u < KS8; ++u) wf[j][u] = __builtin_nontemporal_load((const V8*)(W + ((long)j * (K >> 5) + u) * 512));
    // this wave's slice of the activation rows: KE / 8 16-byte pieces per row
    constexpr int PPR = KE / 8, NP = (RMAX * PPR + 63) / 64;
    V8 xv[NP];
    float ss[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) ss[r] = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = p * 64 + lane, row = i / PPR, c = i % PPR;
        if (row < R) {
            xv[p] = *(const V8*)((const T*)a.X + (long)row * a.ldx + wk * KE + c * 8);
            if (NORM) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)xv[p][e]; s += f * f; }
#pragma unroll
                for (int r = 0; r < RMAX; ++r) if (r == row) ss[r] += s;
            }
        }
    }
    if (NORM) {
        // row sums of squares: lanes -> wave -> the eight waves' partials in wave order; then w * T(x * rstd) on the staged pieces (modeling_llama.py:60-65)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { const float t = wave_sum(ss[r]); if (lane == 0) part[r][wk] = t; }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = p * 64 + lane, row = i / PPR, c = i % PPR;
            if (row < R) {
                float tot = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) tot += part[row][w8];
                const float rs = 1.0f / sqrtf(tot / (float)K + a.eps);
                const float* nw = a.norm_w + wk * KE + c * 8;
                const f32x4 w0 = *(const f32x4*)nw, w1 = *(const f32x4*)(nw + 4);
                V8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (T)(w0[e] * rT<T>((float)xv[p][e] * rs)); o[4 + e] = (T)(w1[e] * rT<T>((float)xv[p][4 + e] * rs)); }
                xv[p] = o;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int i = p * 64 + lane, row = i / PPR, c = i % PPR;
        if (row < R) *(V8*)(xs + (long)row * K + wk * KE + c * 8) = xv[p];
    }
    // (a wave multiplies only the k-steps of its own eighth, i.e. only what it staged itself: no barrier; the LDS writes above are ordered before the reads below
    //  within the wave by the compiler's lgkmcnt bookkeeping)
    float acc[TPB][RMAX];
#pragma unroll
    for (int j = 0; j < TPB; ++j)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[j][r] = 0.f;
    const int kc = lane >> 4;                                              // this lane's 8 k of a 32-deep step
#pragma unroll
    for (int u = 0; u < KS8; ++u) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            if (r < R) {
                const V8 xr = *(const V8*)(xs + (long)r * K + wk * KE + u * 32 + kc * 8);
#pragma unroll
                for (int j = 0; j < TPB; ++j) {
                    float s = acc[j][r];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        V2 wa, xa;
                        wa[0] = wf[j][u][2 * e]; wa[1] = wf[j][u][2 * e + 1]; xa[0] = xr[2 * e]; xa[1] = xr[2 * e + 1];
                        s = Dot2<T>::f(wa, xa, s);
                    }
                    acc[j][r] = s;
                }
            }
        }
    }
    // the four lanes of an output row (k chunks 0 .. 3), then the eight waves, in fixed order
#pragma unroll
    for (int j = 0; j < TPB; ++j)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            float s = acc[j][r];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16 && r < R) red[((wk * TPB + j) * RMAX + r) * 16 + lane] = s;
        }
    __syncthreads();
    // one thread per output: (tile j, row r, column n of the tile)
    for (int o = tid; o < TPB * RMAX * 16; o += 512) {
        const int j = o / (RMAX * 16), r = (o / 16) % RMAX, n = o % 16;
        if (r >= R) continue;
        float s = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) s += red[((w8 * TPB + j) * RMAX + r) * 16 + n];
        const long col = ((long)blockIdx.x * TPB + j) * 16 + n;
        if (MODE == GEMV_SLAB_NORM) a.P[(long)r * a.N + col] = s;
        else if (MODE == GEMV_RESID) {
            T* xp = (T*)a.resid + (long)r * a.ldr + col;
            *xp = (T)((float)*xp + rT<T>(s));
        } else if (n < 8) {
            // gate column n of the tile and its up partner n + 8 (8-row interleave of launch_tile_weights_gu8): read the partner's eight partials as well
            float up = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) up += red[((w8 * TPB + j) * RMAX + r) * 16 + n + 8];
            ((T*)a.act)[(long)r * (a.N / 2) + ((long)blockIdx.x * TPB + j) * 8 + n] = (T)(rT<T>(silu_f(rT<T>(s))) * rT<T>(up));
        }
    }
}

template <typename T, int MODE, int TPB, int KS8> static void launch_gemv_v(const GemvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)GEMV_MAX_ROWS * KS8 * 256 * sizeof(T) + (size_t)8 * TPB * GEMV_MAX_ROWS * 16 * 4;
    if (lds > 65536) ensure_dyn_lds((const void*)gemv_kernel<T, MODE, TPB, KS8>, (int)lds);
    hipLaunchKernelGGL((gemv_kernel<T, MODE, TPB, KS8>), dim3(a.N / (16 * TPB)), dim3(512), lds, s, a);
}
// shapes: K in {256, 512, 1024, 2048} for every mode, K = 6144 / 768-multiples up to 6144 for the down projection; tiles per block chosen for one block per CU where the tile count allows
bool gemv_eligible(int M, int N, int K, int mode) {
    if (M < 1 || M > GEMV_MAX_ROWS || N % 16) return false;
    return K == 256 || K == 512 || K == 1024 || K == 2048 || (mode == GEMV_RESID && (K == 768 || K == 1536 || K == 3072 || K == 6144));
}
template <typename T, int MODE> static void launch_gemv_m(const GemvArgs& a, hipStream_t s) {
    const int tiles = a.N / 16;
#define GV(KS8) do { if (MODE == GEMV_SWIGLU_NORM && tiles % 3 == 0 && KS8 <= 8) launch_gemv_v<T, MODE, 3, KS8>(a, s); \
                     else if (tiles % 4 == 0 && tiles / 4 >= 512 && KS8 <= 8) launch_gemv_v<T, MODE, 4, KS8>(a, s);     /* the lm_head: 3704 tiles -> 926 blocks */ \
                     else if (MODE == GEMV_SWIGLU_NORM && tiles % 2 == 0 && KS8 <= 8) launch_gemv_v<T, MODE, 2, KS8>(a, s); \
                     else launch_gemv_v<T, MODE, 1, KS8>(a, s); } while (0)
    switch (a.K) {
        case 256: GV(1); break; case 512: GV(2); break; case 1024: GV(4); break; case 2048: GV(8); break;
        case 768: if constexpr (MODE == GEMV_RESID) launch_gemv_v<T, MODE, 1, 3>(a, s); break;
        case 1536: if constexpr (MODE == GEMV_RESID) launch_gemv_v<T, MODE, 1, 6>(a, s); break;
        case 3072: if constexpr (MODE == GEMV_RESID) launch_gemv_v<T, MODE, 1, 12>(a, s); break;
        case 6144: if constexpr (MODE == GEMV_RESID) launch_gemv_v<T, MODE, 1, 24>(a, s); break;
    }
#undef GV
}
void launch_gemv(const GemvArgs& a, int mode, hipStream_t s) {
    DT_SWITCH(a.dt, T, {
        if (mode == GEMV_SLAB_NORM) launch_gemv_m<T, GEMV_SLAB_NORM>(a, s);
        else if (mode == GEMV_RESID) launch_gemv_m<T, GEMV_RESID>(a, s);
        else launch_gemv_m<T, GEMV_SWIGLU_NORM>(a, s);
    });
}
