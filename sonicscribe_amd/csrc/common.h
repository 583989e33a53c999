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
    int gemm256_stagger = 1;   // 256x256 GEMM: SIMD partner waves run half a phase apart
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

// ---- GEMM launch descriptors (gemm.hip) ----
enum GemmEpi {
    EPI_BIAS = 0,       // C = bf16(acc + bias)
    EPI_BIAS_GELU = 1,  // C = bf16(gelu(bf16(acc + bias)))
    EPI_BIAS_RESID = 2, // C = bf16(bf16(acc + bias) + R)
    EPI_SWIGLU = 3,     // rows of W interleaved gate/up in 16-row groups: C[:, n/2] = bf16(bf16(silu(bf16 g)) * bf16 u)
    EPI_QKV_VT = 4,     // encoder QKV: columns < n_split -> C (row-major, ldc); columns >= n_split -> V^T [seg][col][t]
};

struct GemmArgs {
    const bf16_t* A; long lda;       // [M][K] row stride lda (elements); lda < K allowed (overlapping im2col rows)
    const bf16_t* W;                 // [N][K] contiguous rows (torch Linear layout)
    bf16_t* C; long ldc;
    const float* bias;               // [N] or null
    const bf16_t* R; long ldr;       // residual (EPI_BIAS_RESID)
    int M, N, K;
    int batch; long strideA, strideC, strideR;   // blockIdx.z batches (conv stem: one per segment)
    // EPI_QKV_VT
    bf16_t* Vt; int n_split; int seg_T; int vt_ld; long vt_seg_stride;  // Vt[seg][n - n_split][t], row stride vt_ld
};

struct SkinnyArgs {
    const bf16_t* X; long ldx;       // [M<=64][K]
    const bf16_t* W;                 // [N][K] in fragment-tiled order (launch_tile_weights)
    float* P;                        // partial slabs [ksplit][Mpad][N] fp32
    int M, N, K, ksplit;
};

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
void launch_skinny(const SkinnyArgs& a, hipStream_t s);
int skinny_pick_ksplit(int N, int K);
bool skinny_gu_eligible(int M, int N, int K);
void launch_skinny_gu(const SkinnyArgs& a, bf16_t* act, hipStream_t s);
void launch_skinny_gu_norm(const SkinnyArgs& a, bf16_t* act, const float* SS, int nblk, const float* w, float eps, hipStream_t s);
bool skinny_o_eligible(int M, int N, int K);
void launch_skinny_o(const SkinnyArgs& a, bf16_t* x, int ldxres, float* SS, hipStream_t s);
void launch_tile_weights(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t s);
void launch_tile_weights_gu8(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t s);   // gate/up: 16-row interleaved source -> 8-row interleaved tiles
