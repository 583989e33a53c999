// Shared device/host helpers for the sonic_hip engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// Activation / weight element type of an engine: bf16 (mode "native", asr.py:61) or IEEE half (mode "int8": the reference runs its
// non-quantised ops in torch.float16, asr.py:61,296).  Buffers are typed bf16_t* throughout the host code (2-byte storage); kernels
// are templated on T and reinterpret.  DT_* is the runtime tag the launchers switch on.
enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };   // DT_F32: only the greedy controller is instantiated for it (SONIC_MODE_F32, f32kind.hip)
template <typename T> struct ET;
template <> struct ET<bf16_t> {
    typedef bf16x8 v8; typedef bf16x4 v4; typedef bf16x2 v2;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct ET<f16_t> {
    typedef f16x8 v8; typedef f16x4 v4; typedef f16x2 v2;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <> struct ET<float> { typedef f32x8 v8; typedef f32x4 v4; };
template <typename T> __device__ __forceinline__ float rT(float x) { return (float)((T)x); }   // round-trip through the element type (RNE)
// fp16: the fp32 value must exist before it is rounded to half.  Without the barrier hipcc folds `fma -> f16` into one
// v_fma_mixlo_f16 (a single rounding of the exact result); torch / CUDA round the op's fp32 result and then convert (two roundings),
// and the two disagree on exact ties (measured: 4e-5 of the dequantised outputs of an int8 GEMM).
template <> __device__ __forceinline__ float rT<f16_t>(float x) { asm volatile("" : "+v"(x)); return (float)((f16_t)x); }
// host-side dispatch on the runtime tag: DT_SWITCH(dt, T, launch<T>(...))
#define DT_SWITCH(dt, T, ...) do { if ((dt) == DT_F16) { typedef f16_t T; __VA_ARGS__; } else { typedef bf16_t T; __VA_ARGS__; } } while (0)

#define WAVE 64

// Opt-in to > 64 KiB of dynamic LDS for a kernel.  The attribute is per DEVICE, so the "done" set is keyed by (device, kernel):
// a second engine on another GPU of the same process gets its own opt-in (a process-wide flag once left device 1 without it).
#include <mutex>
#include <set>
#include <utility>
inline void ensure_dyn_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({dev, fn}).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Experiment knobs.  They belong to an ENGINE (sonic_set_option stores them there); the launchers read this thread-local copy,
// which every locked C-ABI entry point refreshes from its engine before it launches anything.
struct LaunchOpts {
    int skinny_variant = 0;    // 0: shared-X kernel where the shape allows, else one-shot; 2: one-shot nt; 3: one-shot plain loads; 9: read floor (bench only)
    int gemm_force128 = 0;     // route every GEMM to the 128x128 kernel
    int no_fused_gu = 0;       // decode: unfused o_proj / add+RMSNorm / gate-up path
    int no_fused_gu64 = 0;     // ... only for batches of 33 .. 64 rows (round 3's path there; A/B)
    int ktrace_wave = 0;       // in-kernel timeline of skinny_gu64_kernel: which wave stamps the inner points
    int gu64_split_norm = 0;   // 33 .. 64 rows: RMSNorm by its own one-block-per-row kernel (rmsnorm_ss_kernel) + gate/up without the in-LDS norm.  0 = in the chunk graphs of
                               // continuous decode loops only (they share the GPU by design: fewer CU-microseconds beat a shorter chain), 1 = always, -1 = never (A/B).  Same bits.
    int gu64_two_pass = 0;     // fused gate/up at 33 .. 64 rows: round 4's two passes of 32 rows instead of skinny_gu64_kernel (A/B)
    int o64_16rows = 0;        // fused o_proj at 33 .. 64 rows: 16-row blocks (round 4) instead of 32-row ones (A/B)
    int no_skinny48 = 0;       // decode skinny GEMM: never the 48-row x 512 blocks (A/B)
    int no_skinny768 = 0;      // decode skinny GEMM: never the 768-deep K slices (A/B)
    int gemm_small_eff = 75;   // 256x256 grids that under-fill the chip go to the 128x128 kernel, priced at this % of the big kernel's rate (0: never)
    int gemm128_shallow = 0;   // 128x128 GEMM: always the two-stage ring (A/B)
    int no_skinny_i8_wide = 0; // int8 decode skinny GEMM: always 32 rows x 1024 per block (A/B)
    int gemm256_stagger = 1;   // 256x256 GEMM: SIMD partner waves run half a phase apart
    int decode_prefetch = 0;   // bit 0: the decode attention launch's idle CUs (<= 32 rows) stream o_proj's weights, bit 1: ... and the first half of gate/up's, bit 2: the
                               // add+RMSNorm launch's idle CUs stream the next q|k|v's (experiment, profiles/round6_prefetch_ab.txt)
    int decode_attn_occ2 = 0;  // decode attention compiled for 128 VGPRs (two 8-wave blocks per CU can co-reside; a few spilled registers): A/B
    int decode_attn_v1 = 0;    // decode attention with P.V on the VALU (round 2), for A/B runs
    int flash_variant = 2;     // prefill / encoder attention: bit 0 two LDS buffers + one barrier per tile (no gain measured), bit 1 lazy accumulator rescale (-0.6 ms per batch; default)
    int flash_enc = 1;         // encoder attention (head dim 64, no mask): 0 = flash_attn_kernel (rounds 1-4), v > 0 = flash_enc_kernel mode v - 1 (attn_enc.hip; round 5)
    int gemm256_persist = 0;   // 1: 16-bit 256x256 GEMM as one persistent launch (gemm256p.hip) - measured SLOWER than one block per tile (round 3), kept for A/B
    int gemm256_persist_cus = 256;   // ... on at most this many CUs (the rest stay free for whatever else runs; A/B)
    int gemm256_gm = 8;        // 256x256 GEMM raster: M tiles per group (a group sweeps all N tiles before the next M rows)
};
extern thread_local LaunchOpts g_opts;

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }          // v_cvt_pk_bf16_f32: RNE, NaN-safe
__device__ __forceinline__ float rbf(float x) { return (float)((bf16_t)x); }  // round-trip through bf16

// exact-form GELU 0.5*x*(1+erf(x/sqrt2)) with erf from Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7): a dozen VALU ops
// instead of libm erff (which cost ~25 % of the fc1 GEMM in its epilogue).  For x < 0 the complementary form
// 1 + erf(x) = poly(t) * exp(-z^2) is used directly, so there is no cancellation in the tail.  The result is rounded to
// bf16 by the caller, which is 3 orders of magnitude coarser than the approximation error.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float pe = p * t * __expf(-z * z);          // = 1 - erf(z) = erfc(z)
    return 0.5f * x * (x >= 0.f ? 2.0f - pe : pe);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Idle-CU weight prefetch (round 6 experiment, option decode_prefetch): blocks beyond a kernel's own grid stream a byte range - the weights a LATER kernel of
// the token step will read - through plain (temporal) loads whose data is discarded, so the range is resident in the Infinity Cache / an XCD's L2 when its
// consumer asks for it.  Block `blk` of `nblk` takes a contiguous share; 16 bytes per lane and load, every load in flight before the one wait.
struct PrefetchRange { const void* p; long bytes; };
__device__ __forceinline__ void prefetch_share(const PrefetchRange& r, int blk, int nblk) {
    if (!r.p || r.bytes <= 0) return;
    const long step = (long)blockDim.x * 16;
    const long per = ((r.bytes + nblk - 1) / nblk + step - 1) / step * step;
    const long lo = (long)blk * per, hi = lo + per < r.bytes ? lo + per : r.bytes;
    const char* q = (const char*)r.p;
    // the destination registers stay allocated ("+v") until the loads have landed: an output the compiler believes dead would be handed to the address
    // arithmetic of the next iteration while the data of an earlier load is still on its way into it
    i32x4 t = {0, 0, 0, 0};
    for (long off = lo + (long)threadIdx.x * 16; off + 16 <= hi; off += step)
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(t) : "v"(q + off) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t) :: "memory");
}

// ---- GEMM launch descriptors (gemm.hip) ----
enum GemmEpi {
    EPI_BIAS = 0,       // C = bf16(acc + bias)
    EPI_BIAS_GELU = 1,  // C = bf16(gelu(bf16(acc + bias)))
    EPI_BIAS_RESID = 2, // C = bf16(bf16(acc + bias) + R)
    EPI_SWIGLU = 3,     // rows of W interleaved gate/up in 16-row groups: C[:, n/2] = bf16(bf16(silu(bf16 g)) * bf16 u)
    EPI_QKV_VT = 4,     // encoder QKV: columns < n_split -> C (row-major, ldc); columns >= n_split -> V^T [seg][col][t]
};

// int8 GEMM (Linear8bitLt): A and W are int8, the epilogue dequantises with the row statistics and adds the outlier columns
struct GemmI8 {
    const float* sca;                 // [M] row absmax of the quantised activations (NULL: not an int8 GEMM)
    const float* scb;                 // [N] row absmax of the weights
    const bf16_t* x16; long ldx16;    // unquantised activations (fp16 storage) for the outlier columns
    const int* oc_cnt; const int* oc_list; int oc_ld;   // outlier columns per group
    const int* row_group; int group_div;                 // group of row m = row_group ? row_group[(m + row_off) / group_div] : (m + row_off) / group_div
    int row_off;                                         // index of this launch's row 0 in the quantised matrix (a launch over a row range)
    // Deferred outliers (EPI_BIAS_RESID only): rows whose group lists more than defer_thr outlier columns leave the GEMM as fp16(acc * s + b)
    // in defer_out (same row pitch as C; no outlier sum, no residual) and launch_i8_outlier_side finishes them with a dense fp16 MFMA
    // product over the gathered columns.  NULL: every list is walked in the epilogue (O(columns) scalar loads per output element).
    bf16_t* defer_out; int defer_thr;
    // (rounds 3 - 5 had `wk` here: a k-major copy of W for the outlier columns of a tiled operand.  Round 6 gathers them from the tiled copy itself - i8_tiled_row_off /
    //  i8_tiled_k_off with GemmArgs.w_tiled - and the copy is gone: 1.29 GB at full size)
};

// byte offset of W[n][k] inside the fragment-tiled int8 copy (launch_tile_weights_i8: (16-row, 64-k) tiles of 1 KiB): the row part and the k part add
__host__ __device__ __forceinline__ long i8_tiled_row_off(int n, int K) { return (long)(n >> 4) * (K >> 6) * 1024 + (long)(n & 15) * 16; }   // consecutive n of a 16-row group: 16 bytes apart
__host__ __device__ __forceinline__ long i8_tiled_k_off(int k) { return ((long)(k >> 6) << 10) + (((k >> 4) & 3) << 8) + (k & 15); }

struct GemmArgs {
    const bf16_t* A; long lda;       // [M][K] row stride lda (elements); lda < K allowed (overlapping im2col rows).  int8 GEMM: int8_t data
    const bf16_t* W;                 // [N][K] contiguous rows (torch Linear layout).  int8 GEMM: int8_t data
    bf16_t* C; long ldc;
    const float* bias;               // [N] or null
    const bf16_t* R; long ldr;       // residual (EPI_BIAS_RESID)
    int M, N, K;
    int batch; long strideA, strideC, strideR;   // blockIdx.z batches (conv stem: one per segment)
    // EPI_QKV_VT
    bf16_t* Vt; int n_split; int seg_T; int vt_ld; long vt_seg_stride;  // Vt[seg][n - n_split][t], row stride vt_ld
    int dt;                          // DT_BF16 / DT_F16: element type of A, W (16-bit GEMM) and of C, R, Vt
    GemmI8 q;                        // q.sca != NULL: int8 GEMM, outputs fp16
    // fused encoder RoPE (256x256 kernel, heads of 64 columns, rotary dim 32: modeling_glmasr.py:153-168): columns < rope_ncols are
    // q / k heads whose first 32 dims are rotated in the epilogue, row m sits at position m % rope_T; table [rope_T][32] = cos | sin
    const float* rope_cs; int rope_T, rope_ncols;
    // bf16 GELU by table (256x256 kernel, EPI_BIAS_GELU): GELU of a bf16 value is a function of 65536 inputs; the compact table
    // (GELU_LUT_N bf16 bit patterns: both signs x exponents 2^-14 .. 2^3 x 128 mantissas) is copied to LDS and indexed by the bits
    const unsigned short* gelu_lut;
    int raster_gm;                   // 256x256 kernel: M tiles per raster group (0: default 8)
    int w_tiled;                     // 16-bit kinds: W is the decode step's fragment-tiled copy ([N / 16][K / 32][64 lanes][8], launch_tile_weights) - the ONE copy of a
                                     // decoder projection since round 5; a 1-KiB LDS-DMA piece is then one (16-row, 32-k) fragment and the MFMA operand read is lane-linear
    int gu8;                         // EPI_SWIGLU: gate / up rows interleaved in 8-row groups (launch_tile_weights_gu8) instead of 16-row groups
    long long* dbg;                  // diagnostics (sonic_bench_gemm with option gemm_trace): per block 8 words {entry, first K tile landed, K loop done, stores issued (100 MHz clock), hw id}; null in production
};
#define GELU_LUT_E0 113                     // biased exponent of 2^-14
#define GELU_LUT_NE 18                      // exponents 2^-14 .. 2^3  (|x| < 16)
#define GELU_LUT_HALF (GELU_LUT_NE * 128)
#define GELU_LUT_N (2 * GELU_LUT_HALF)      // 4608 entries = 9216 bytes
// GELU of a bf16 value l by the table (LDS copy in the 256x256 kernel, the global copy in the 128x128 one: both kernels must give the same
// bits, or a request's result would depend on which kernel its batch size selects).  Branch-free: a clamped read + two selects.
__device__ __forceinline__ int gelu_lut_index(float l) {
    return (int)((__float_as_uint(l) >> 16) & 0x7FFFu) - (GELU_LUT_E0 << 7);
}
__device__ __forceinline__ int gelu_lut_slot(float l, int idx) {
    return (int)min((unsigned)idx, (unsigned)(GELU_LUT_HALF - 1)) + ((__float_as_uint(l) >> 31) ? GELU_LUT_HALF : 0);
}
__device__ __forceinline__ float gelu_lut_value(float l, int idx, unsigned t) {
    // outside the table: |x| < 2^-14 -> 0.5 x (the erf term is below half a bf16 ulp); |x| >= 16 -> x or -0
    const float lo = 0.5f * l, hi = fmaxf(l, -0.0f);
    float y = __uint_as_float(t << 16);
    y = idx < 0 ? lo : y;
    y = idx >= GELU_LUT_HALF ? hi : y;
    return y;
}
bool gemm256_eligible(const GemmArgs& a, int epi);

// Operand kinds of the MFMA GEMM kernels: element, 16-byte fragment, accumulator, output element type
struct KBF16 {
    typedef bf16_t elem; typedef bf16_t out; typedef bf16x8 frag; typedef f32x4 acc; static constexpr bool I8 = false;
    static __device__ __forceinline__ acc mfma(frag a, frag b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
struct KF16 {
    typedef f16_t elem; typedef f16_t out; typedef f16x8 frag; typedef f32x4 acc; static constexpr bool I8 = false;
    static __device__ __forceinline__ acc mfma(frag a, frag b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
struct KI8 {   // v_mfma_i32_16x16x64_i8: 16 int8 of k per lane and step, lane l: row l & 15, k = 16 * (l >> 4) + j
    typedef int8_t elem; typedef f16_t out; typedef i32x4 frag; typedef i32x4 acc; static constexpr bool I8 = true;
    static __device__ __forceinline__ acc mfma(frag a, frag b, acc c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }
};
// host-side dispatch: KD_SWITCH(args, KD, launch<KD>(...))
#define KD_SWITCH(a, KD, ...) do { if ((a).q.sca) { typedef KI8 KD; __VA_ARGS__; } else if ((a).dt == DT_F16) { typedef KF16 KD; __VA_ARGS__; } \
                                   else { typedef KBF16 KD; __VA_ARGS__; } } while (0)

struct SkinnyArgs {
    const bf16_t* X; long ldx;       // [M<=64][K]   (i8: int8_t data)
    const bf16_t* W;                 // [N][K] in fragment-tiled order (launch_tile_weights / launch_tile_weights_i8)
    float* P;                        // partial slabs [ksplit][Mpad][N] fp32 (i8: int32 bit patterns)
    int M, N, K, ksplit;
    int dt;                          // DT_BF16 / DT_F16
    int i8;                          // int8 operands (Linear8bitLt decode step): int32 slabs, dequantised by the consumer
    const float* x_amax;             // i8 only, optional: X holds the UNQUANTISED fp16 rows and x_amax[row][0..3] partial maxima (written by the
                                     // producer's blocks) of their absmax without the elements >= 6.0: the kernel quantises its X slice while
                                     // staging it - int8 = rn(x * 127 / absmax), 0 for outliers (LLM.int8 row-wise)
    // PRE form (skinny_xs_kernel<.., PRE>, M <= 2): X is not read - the block computes X = RMSNorm(pre_x + sum of the pre_ks slabs pre_P[ks][pre_mpad][K]) * pre_w
    // itself (add_rmsnorm_kernel's arithmetic); block (0, 0) writes the updated residual rows to pre_xout (a buffer other than pre_x; may be null)
    const float* pre_P; int pre_ks, pre_mpad; const bf16_t* pre_x; bf16_t* pre_xout; const float* pre_w; float pre_eps;
    int* err;                        // optional device error word (engine: n_active[1]): a kernel that gives up on an in-kernel wait ORs 1 into it; the greedy
                                     // kernel turns it into a negative running-row count, which every host-side check reads as a failed step
    long long* kt;                   // diagnostics: per-block timestamps [block][8] (100 MHz wall clock), null in production
    int kt_thread;                   // ... the thread that stamps the inner points of skinny_gu64_kernel (0, 64 .. 448: one wave's view each; option ktrace_wave)
};
// in-kernel timeline point `slot` of this block (thread 0 only); a null pointer costs one scalar compare
// (the pointer is laundered through an SGPR so that the address arithmetic stays inside the branch: hoisted, it cost the 256-register kernels a spill)
// and the index is 32-bit scalar arithmetic - the 64-bit form went through v_mad_u64_u32 on a spilled operand, i.e. a scratch reload + vmcnt(0) at every point)
#define KT(a, slot) do { if ((a).kt && threadIdx.x == 0) { long long* kt_p_ = (a).kt; asm volatile("" : "+s"(kt_p_)); \
        const unsigned kt_i_ = __builtin_amdgcn_readfirstlane((blockIdx.y * gridDim.x + blockIdx.x) * 8u + (unsigned)(slot)); \
        kt_p_[kt_i_] = wall_clock64(); } } while (0)
#define KTW(a, slot) do { if ((a).kt && threadIdx.x == (unsigned)(a).kt_thread) { long long* kt_p_ = (a).kt; asm volatile("" : "+s"(kt_p_)); \
        const unsigned kt_i_ = __builtin_amdgcn_readfirstlane((blockIdx.y * gridDim.x + blockIdx.x) * 8u + (unsigned)(slot)); \
        kt_p_[kt_i_] = wall_clock64(); } } while (0)

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
void launch_skinny(const SkinnyArgs& a, hipStream_t s);
int skinny_pick_ksplit(int N, int K);
bool skinny_pre_eligible(int M, int N, int K);
int skinny_pick_ksplit_i8(int N, int K);
bool skinny_gu_eligible(int M, int N, int K);
void launch_skinny_gu(const SkinnyArgs& a, bf16_t* act, hipStream_t s);
void launch_skinny_gu_norm(const SkinnyArgs& a, bf16_t* act, const float* SS, int nblk, const float* w, float eps, hipStream_t s);
bool skinny_o_eligible(int M, int N, int K);
void launch_skinny_o(const SkinnyArgs& a, bf16_t* x, int ldxres, float* SS, hipStream_t s);
void launch_tile_weights(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t s);
void launch_tile_weights_gu8(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t s);   // gate/up: 16-row interleaved source -> 8-row interleaved tiles
