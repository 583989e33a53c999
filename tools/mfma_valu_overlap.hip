// Do MFMA and VALU work overlap on a gfx950 SIMD - across waves, and inside one wave?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap && tools/mfma_valu_overlap
// One block per CU, 512 threads = 2 waves per SIMD.  mode 0: every wave runs a chain of v_mfma_f32_16x16x32_bf16 (4 independent accumulators);
// 1: every wave runs VALU work (fma / exp2 mix of a softmax: 4 fma + 1 exp per group); 2: even waves MFMA, odd waves VALU (the two kinds
// share every SIMD); 3: every wave runs both, interleaved in its own instruction stream.  Times per kernel; ideal overlap = max, none = sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wid = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, e = 0.f;
    const bool do_m = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4) && !(wid & 1));
    const bool do_v = MODE == 1 || MODE == 3 || ((MODE == 2 || MODE == 5) && (wid & 1));
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x0 * 1e-6f);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x1 * 1e-6f);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                x0 = __builtin_fmaf(x0, 0.9999f, 0.25f); x1 = __builtin_fmaf(x1, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x2 * 1e-6f);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                x2 = __builtin_fmaf(x2, 0.9999f, 0.25f); x3 = __builtin_fmaf(x3, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x3 * 1e-6f);
            }
        }
    } else {
        if (do_m)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                }
            }
        if (do_v)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x0 * 1e-6f);
                    x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x1 * 1e-6f);
                    x0 = __builtin_fmaf(x0, 0.9999f, 0.25f); x1 = __builtin_fmaf(x1, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x2 * 1e-6f);
                    x2 = __builtin_fmaf(x2, 0.9999f, 0.25f); x3 = __builtin_fmaf(x3, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x3 * 1e-6f);
                }
            }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + e;
}
template <int MODE>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    const int wid = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 c0 = {0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, e = 0.f;
    const bool do_m = MODE == 0 || MODE == 3 || ((MODE == 2 || MODE == 4) && !(wid & 1));
    const bool do_v = MODE == 1 || MODE == 3 || ((MODE == 2 || MODE == 5) && (wid & 1));
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x0 * 1e-6f);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x1 * 1e-6f);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                x0 = __builtin_fmaf(x0, 0.9999f, 0.25f); x1 = __builtin_fmaf(x1, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x2 * 1e-6f);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
                x2 = __builtin_fmaf(x2, 0.9999f, 0.25f); x3 = __builtin_fmaf(x3, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x3 * 1e-6f);
            }
        }
    } else {
        if (do_m)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
                }
            }
        if (do_v)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x0 * 1e-6f);
                    x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f); e += __builtin_amdgcn_exp2f(x1 * 1e-6f);
                    x0 = __builtin_fmaf(x0, 0.9999f, 0.25f); x1 = __builtin_fmaf(x1, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x2 * 1e-6f);
                    x2 = __builtin_fmaf(x2, 0.9999f, 0.25f); x3 = __builtin_fmaf(x3, 0.9999f, 0.25f); e += __builtin_amdgcn_exp2f(x3 * 1e-6f);
                }
            }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + e;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](const char* name, auto kern, double mfma_per_wave, double valu_per_wave) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, 100); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s %8.3f ms   per wave: %.0f MFMA, %.0f VALU-op groups\n", name, ms, mfma_per_wave, valu_per_wave);
    };
    run("all 8 waves: MFMA only", k<0>, 16.0 * iters, 0);
    run("all 8 waves: VALU only (8 fma/mul + 4 exp / 4 MFMA slots)", k<1>, 0, 16.0 * iters);
    run("4 waves MFMA + 4 waves VALU (one of each per SIMD)", k<2>, 16.0 * iters, 16.0 * iters);
    run("all 8 waves: both, interleaved in the wave", k<3>, 16.0 * iters, 16.0 * iters);
    run("4 waves MFMA, the other 4 idle", k<4>, 16.0 * iters, 0);
    run("4 waves VALU, the other 4 idle", k<5>, 0, 16.0 * iters);
    printf("the same with v_mfma_f32_32x32x16_bf16 (twice the FLOPs per instruction):\n");
    run("all 8 waves: MFMA only", k32<0>, 16.0 * iters, 0);
    run("all 8 waves: VALU only", k32<1>, 0, 16.0 * iters);
    run("4 waves MFMA + 4 waves VALU (one of each per SIMD)", k32<2>, 16.0 * iters, 16.0 * iters);
    run("all 8 waves: both, interleaved in the wave", k32<3>, 16.0 * iters, 16.0 * iters);
    run("4 waves MFMA, the other 4 idle", k32<4>, 16.0 * iters, 0);
    run("4 waves VALU, the other 4 idle", k32<5>, 0, 16.0 * iters);
    return 0;
}
