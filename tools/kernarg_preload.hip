// Does kernel-argument preloading (-mllvm -amdgpu-kernarg-preload-count=N: the first N dwords of explicit kernel arguments arrive in
// SGPRs with the wave, no s_load round trip) shorten the head of a short decode-step kernel?  A hipGraph chain of 170 dependent small
// kernels (each block reads 16 KiB behind a pointer taken from its arguments), arguments passed (a) as one by-value
// struct (what the engine's kernels did: byref arguments cannot be preloaded) and (b) as leading scalar arguments.  Build twice:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/kp0 tools/kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/kp1 tools/kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

struct Args { const v4i* p; int* out; long stride16; int n; int pad[9]; };

__device__ __forceinline__ void body(const v4i* p, int* out, long stride16, int n) {
    const v4i* q = p + (long)blockIdx.x * stride16 + threadIdx.x;
    v4i a = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) a ^= __builtin_nontemporal_load(q + u * 512);
    if ((a[0] ^ a[1] ^ a[2] ^ a[3] ^ n) == 0x5a17c0de) out[blockIdx.x] = 1;      // every lane's loads stay live
}
__global__ __launch_bounds__(512) void k_struct(Args a) { body(a.p, a.out, a.stride16, a.n); }
__global__ __launch_bounds__(512) void k_scalar(const v4i* p, int* out, long stride16, int n) { body(p, out, stride16, n); }

int main() {
    const int NK = 170, BLK = 256;
    const size_t bytes = (size_t)BLK * 16384;
    v4i* buf; int* out;
    CK(hipMalloc(&buf, bytes * NK)); CK(hipMemset(buf, 1, bytes * NK)); CK(hipMalloc(&out, BLK * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int variant = 0; variant < 2; ++variant) {
        hipGraph_t g; hipGraphExec_t gx;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NK; ++i) {
            const v4i* p = buf + (size_t)i * (bytes / 16);
            if (variant == 0) { Args a{p, out, 1024, i, {0}}; hipLaunchKernelGGL(k_struct, dim3(BLK), dim3(512), 0, s, a); }
            else hipLaunchKernelGGL(k_scalar, dim3(BLK), dim3(512), 0, s, p, out, 1024L, i);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
        for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(gx, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int R = 200;
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(gx, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f us per kernel (chain of %d, %d replays)\n", variant == 0 ? "by-value struct " : "scalar arguments", ms * 1e3 / R / NK, NK, R);
        CK(hipGraphExecDestroy(gx)); CK(hipGraphDestroy(g));
    }
    return 0;
}
