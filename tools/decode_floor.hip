// What does the LAUNCH STRUCTURE of the decode token step cost, before any arithmetic?  A hipGraph of pure read kernels with the byte
// counts of the real step (per layer: qkv 12.6 MB, KV cache, o_proj 8.4 MB, gate/up 50.3 MB, down 25.2 MB, a tiny add+RMSNorm; then the
// 243 MB lm_head and a tiny greedy kernel), 28 layers, replayed like the engine replays its captured step.  Variants: threads per
// block, blocks, loads in flight per lane, nontemporal loads, uncached allocation, and a chain of empty kernels (the pure boundary
// cost).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/decode_floor tools/decode_floor.hip && /tmp/decode_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

// every block reads its contiguous share of [p, p + n16 * 16): U 16-byte loads in flight per lane, each wave-instruction 1 KiB
template <int U, bool NT>
__global__ void read_kernel(const v4i* __restrict__ p, long n16, int* sink) {
    const long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    v4i acc = {0, 0, 0, 0};
    const int T = blockDim.x;
    for (long i = lo + threadIdx.x; i < hi; i += (long)T * U) {
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + (long)u * T;
            if (j < hi) v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j]; else v[u] = (v4i){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x5a17c0de) sink[0] = 1;
}
__global__ void empty_kernel(int* sink) { if (threadIdx.x == 4096) sink[0] = 1; }
__global__ void tiny_kernel(const float* in, float* out, int n) {      // stands for add+RMSNorm / greedy: a few KB in, a few KB out
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.0001f;
}

struct Cfg { int threads, blocks, U; bool nt, uc; int lds; };

template <int U> static void launch_read(const Cfg& c, const void* p, size_t bytes, int blocks, int* sink, hipStream_t s) {
    if (c.nt) hipLaunchKernelGGL((read_kernel<U, true>), dim3(blocks), dim3(c.threads), c.lds, s, (const v4i*)p, (long)(bytes / 16), sink);
    else hipLaunchKernelGGL((read_kernel<U, false>), dim3(blocks), dim3(c.threads), c.lds, s, (const v4i*)p, (long)(bytes / 16), sink);
}
static void launch_read_u(const Cfg& c, const void* p, size_t bytes, int blocks, int* sink, hipStream_t s) {
    switch (c.U) { case 4: launch_read<4>(c, p, bytes, blocks, sink, s); break; case 8: launch_read<8>(c, p, bytes, blocks, sink, s); break;
                   case 16: launch_read<16>(c, p, bytes, blocks, sink, s); break; default: launch_read<24>(c, p, bytes, blocks, sink, s); break; }
}

int main(int argc, char** argv) {
    const int L = 28, B = 32, ctx = 335;
    const size_t qkv = 3072ul * 2048 * 2, wo = 2048ul * 2048 * 2, gu = 2ul * 6144 * 2048 * 2, dn = 2048ul * 6144 * 2, head = 59264ul * 2048 * 2;
    const size_t kv = (size_t)B * 4 * ctx * 128 * 2 * 2;       // K and V of one layer
    const size_t per_layer = qkv + wo + gu + dn + kv;
    int* sink; CK(hipMalloc(&sink, 4096));
    float *xa, *xb; CK(hipMalloc(&xa, 1 << 20)); CK(hipMalloc(&xb, 1 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto alloc = [&](void** p, size_t n, bool uc) {
        if (uc) CK(hipExtMallocWithFlags(p, n, hipDeviceMallocUncached)); else CK(hipMalloc(p, n));
        CK(hipMemsetAsync(*p, 0x11, n, s));
    };
    std::vector<Cfg> cfgs = {
        {512, 256, 8, true, true, 0}, {512, 256, 24, true, true, 0}, {512, 256, 24, true, true, 131072}, {256, 256, 24, true, true, 0}, {1024, 256, 8, true, true, 0},
        {256, 512, 16, true, true, 0}, {256, 1024, 8, true, true, 0}, {512, 256, 24, false, true, 0}, {512, 256, 24, true, false, 0}, {512, 256, 24, false, false, 0},
    };
    for (int pass = 0; pass < 2; ++pass) {
        const bool uc = pass == 0;
        std::vector<void*> W(L); void* H;
        for (int l = 0; l < L; ++l) alloc(&W[l], per_layer, uc);
        alloc(&H, head, uc);
        CK(hipStreamSynchronize(s));
        for (const Cfg& c : cfgs) {
            if (c.uc != uc) continue;
            for (int mode = 0; mode < 3; ++mode) {     // 0: the step's kernels as pure reads; 1: empty kernels, same count; 2: one read kernel per layer
                if (mode == 1 && !(c.threads == 512 && c.U == 24 && c.nt)) continue;
                if (c.lds > 65536) {
                    CK(hipFuncSetAttribute((const void*)read_kernel<24, true>, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
                }
                hipGraph_t g; hipGraphExec_t gx;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int l = 0; l < L; ++l) {
                    char* p = (char*)W[l];
                    if (mode == 0) {
                        launch_read_u(c, p, qkv, c.blocks, sink, s); p += qkv;
                        launch_read_u(c, p, kv, 128, sink, s); p += kv;
                        launch_read_u(c, p, wo, c.blocks, sink, s); p += wo;
                        launch_read_u(c, p, gu, c.blocks, sink, s); p += gu;
                        launch_read_u(c, p, dn, c.blocks, sink, s);
                        hipLaunchKernelGGL(tiny_kernel, dim3(32), dim3(512), 0, s, xa, xb, 16384);
                    } else if (mode == 1) {
                        for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(empty_kernel, dim3(c.blocks), dim3(c.threads), c.lds, s, sink);
                    } else {
                        launch_read_u(c, p, per_layer, c.blocks, sink, s);
                    }
                }
                if (mode != 1) launch_read_u(c, H, head, c.blocks, sink, s); else hipLaunchKernelGGL(empty_kernel, dim3(c.blocks), dim3(c.threads), c.lds, s, sink);
                hipLaunchKernelGGL(tiny_kernel, dim3(32), dim3(512), 0, s, xa, xb, 16384);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
                for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(gx, s));
                CK(hipStreamSynchronize(s));
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                const int reps = 40;
                CK(hipEventRecord(a, s));
                for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(gx, s));
                CK(hipEventRecord(b, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                const double us = ms * 1e3 / reps, bytes = (double)per_layer * L + head;
                printf("%-22s threads %4d blocks %4d U %2d nt %d uc %d lds %6d : %8.1f us per step", mode == 0 ? "step kernels as reads" : mode == 1 ? "empty kernels (170)" : "one read per layer", c.threads,
                       c.blocks, c.U, (int)c.nt, (int)c.uc, c.lds, us);
                if (mode != 1) printf("  = %.2f TB/s of %.2f GB", bytes / us / 1e6, bytes / 1e9);
                printf("\n"); fflush(stdout);
                CK(hipGraphExecDestroy(gx)); CK(hipGraphDestroy(g)); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
            }
        }
        for (int l = 0; l < L; ++l) CK(hipFree(W[l]));
        CK(hipFree(H));
    }
    return 0;
}
