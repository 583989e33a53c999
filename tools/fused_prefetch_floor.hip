// Is one persistent launch [o_proj -> grid barrier -> gate/up] faster than the two launches it replaces, when the second op's
// weights are requested (into registers) BEFORE the barrier?  Pure-read stand-ins with the decode step's byte counts:
//   op A (o_proj):   8.4 MB of weights, every block also reads a 64 KB activation image (L2) and publishes 512 B
//   op B (gate/up):  50.3 MB of weights, every block also reads the 128 KB activation image op A published
// Variant 0: two kernels per pair (stream order).  Variant 1: one kernel, 256 blocks (one per CU), B's 24 x 16-byte loads per lane
// issued first, then A's loads; A consumed and published; XCD-style hierarchical grid barrier (group = blockIdx & 7, bounded spin);
// the activation image read behind an agent-scope acquire; B consumed.  (vmcnt retires in order, so the drain of A's stores in front of
// the barrier also waits for every B load: no overlap.)  Variant 3: the form that CAN overlap on gfx9 - 9 waves per block, waves 0-7
// do A, drain their own stores, THEN request B (in-order vmcnt: the store drain must not sit behind B's loads), a ninth wave that never
// has a load in flight arrives at the grid barrier, polls it and runs the acquire (a polling wave's loads would queue behind its own
// B requests).  A chain of 28 pairs in one hipGraph, like the engine's step.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fpf tools/fused_prefetch_floor.hip && /tmp/fpf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

struct Bar { unsigned cnt[8][32]; unsigned top[32]; unsigned gen[32]; unsigned err[32]; };   // every word on a line of its own

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all blocks of the grid (gridDim.x % 8 == 0, all co-resident) meet; data stored before is visible to plain loads after
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned& g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned grp = blockIdx.x & 7, per = gridDim.x >> 3;
        if (__hip_atomic_fetch_add(&b->cnt[grp][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per - 1) {
            __hip_atomic_store(&b->cnt[grp][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7) {
                __hip_atomic_store(&b->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&b->gen[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        int spins = 0;
        while (ld_sc1(&b->gen[0]) == g) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    g += 1;
    __syncthreads();
}

template <int U> __device__ __forceinline__ void issue(const v4i* p, v4i (&v)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + u * 512);
}
template <int U> __device__ __forceinline__ v4i fold(const v4i (&v)[U]) {
    v4i a = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) a ^= v[u];
    return a;
}

// op as its own kernel: U loads per lane of weights + XU loads per lane of the shared activation image, publishes 4 B per lane
template <int U, int XU> __global__ __launch_bounds__(512) void op_kernel(const v4i* w, const v4i* x, int* out, int* sink) {
    v4i wv[U], xv[XU];
    issue<XU>(x + threadIdx.x, xv);
    issue<U>(w + (long)blockIdx.x * U * 512 + threadIdx.x, wv);
    const v4i a = fold<U>(wv) ^ fold<XU>(xv);
    out[blockIdx.x * 512 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
    if (a[0] == 0x5a17c0de) sink[0] = 1;
}

// the pair fused: B's weights requested first, A done, barrier, activation image, B done
template <int UA, int XA, int UB, int XB> __global__ __launch_bounds__(512) void fused_kernel(const v4i* wa, const v4i* xa, const v4i* wb, int* mid, int* out, Bar* bar, int* sink) {
    unsigned g = 0;
    if (threadIdx.x == 0) g = ld_sc1(&bar->gen[0]);
    v4i bv[UB], av[UA], xv[XA];
    issue<XA>(xa + threadIdx.x, xv);
    issue<UA>(wa + (long)blockIdx.x * UA * 512 + threadIdx.x, av);
    issue<UB>(wb + (long)blockIdx.x * UB * 512 + threadIdx.x, bv);
    const v4i a = fold<UA>(av) ^ fold<XA>(xv);
    mid[blockIdx.x * 512 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
    grid_barrier(bar, g);
    v4i x2[XB];
#pragma unroll
    for (int u = 0; u < XB; ++u) x2[u] = ((const v4i*)mid)[u * 512 + threadIdx.x];       // the image op A published (plain loads behind the acquire)
    const v4i b = fold<UB>(bv) ^ fold<XB>(x2);
    out[blockIdx.x * 512 + threadIdx.x] = b[0] ^ b[1] ^ b[2] ^ b[3];
    if (b[0] == 0x5a17c0de) sink[0] = 1;
}

// control-wave form: waves 0..7 work, wave 8 synchronises
__device__ __forceinline__ void grid_barrier_ctrl(Bar* b, unsigned& g, bool ctrl) {
    __builtin_amdgcn_s_barrier();                     // every working wave has drained its own stores (it waited vmcnt before requesting B)
    if (ctrl) {
        if ((threadIdx.x & 63) == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned grp = blockIdx.x & 7, per = gridDim.x >> 3;
            if (__hip_atomic_fetch_add(&b->cnt[grp][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == per - 1) {
                __hip_atomic_store(&b->cnt[grp][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7) {
                    __hip_atomic_store(&b->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&b->gen[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            int spins = 0;
            while (ld_sc1(&b->gen[0]) == g) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { __hip_atomic_store(&b->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    g += 1;
    __builtin_amdgcn_s_barrier();
}

template <int UA, int XA, int UB, int XB> __global__ __launch_bounds__(576) void fused_ctrl_kernel(const v4i* wa, const v4i* xa, const v4i* wb, int* mid, int* out, Bar* bar, int* sink) {
    const bool ctrl = threadIdx.x >= 512;
    unsigned g = 0;
    if (ctrl) g = ld_sc1(&bar->gen[0]);
    v4i bv[UB];
    if (!ctrl) {
        v4i av[UA], xv[XA];
        issue<XA>(xa + threadIdx.x, xv);
        issue<UA>(wa + (long)blockIdx.x * UA * 512 + threadIdx.x, av);
        const v4i a = fold<UA>(av) ^ fold<XA>(xv);
        mid[blockIdx.x * 512 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have left (nothing else of its is in flight yet)
        issue<UB>(wb + (long)blockIdx.x * UB * 512 + threadIdx.x, bv);
    }
    grid_barrier_ctrl(bar, g, ctrl);
    if (!ctrl) {
        v4i x2[XB];
#pragma unroll
        for (int u = 0; u < XB; ++u) x2[u] = ((const v4i*)mid)[u * 512 + threadIdx.x];
        const v4i b = fold<UB>(bv) ^ fold<XB>(x2);
        out[blockIdx.x * 512 + threadIdx.x] = b[0] ^ b[1] ^ b[2] ^ b[3];
        if (b[0] == 0x5a17c0de) sink[0] = 1;
    }
}

int main() {
    const int L = 28, NB = 256;
    constexpr int UA = 4, XA = 8, UB = 24, XB = 16;           // per lane: A 64 B of weights (32 KB / block) + 64 KB image; B 384 B (192 KB / block) + 128 KB image
    const size_t wa_b = (size_t)NB * UA * 512 * 16, wb_b = (size_t)NB * UB * 512 * 16;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<v4i*> WA(L), WB(L);
    for (int l = 0; l < L; ++l) {
        CK(hipExtMallocWithFlags((void**)&WA[l], wa_b, hipDeviceMallocUncached)); CK(hipExtMallocWithFlags((void**)&WB[l], wb_b, hipDeviceMallocUncached));
        CK(hipMemsetAsync(WA[l], 0x11, wa_b, s)); CK(hipMemsetAsync(WB[l], 0x22, wb_b, s));
    }
    v4i* xin; int *mid, *out, *sink; Bar* bar;
    CK(hipMalloc(&xin, 1 << 20)); CK(hipMemsetAsync(xin, 0x33, 1 << 20, s));
    CK(hipMalloc(&mid, NB * 512 * 4)); CK(hipMalloc(&out, NB * 512 * 4)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&bar, sizeof(Bar)));
    CK(hipMemsetAsync(bar, 0, sizeof(Bar), s)); CK(hipMemsetAsync(mid, 0, NB * 512 * 4, s));
    CK(hipStreamSynchronize(s));
    printf("op A %.1f MB + op B %.1f MB per pair, %d pairs per graph\n", wa_b / 1e6, wb_b / 1e6, L);
    for (int round = 0; round < 2; ++round)
        for (int variant = 0; variant < 4; ++variant) {
            hipGraph_t g; hipGraphExec_t gx;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < L; ++l) {
                if (variant == 0) {
                    hipLaunchKernelGGL((op_kernel<UA, XA>), dim3(NB), dim3(512), 0, s, WA[l], xin, mid, sink);
                    hipLaunchKernelGGL((op_kernel<UB, XB>), dim3(NB), dim3(512), 0, s, WB[l], (const v4i*)mid, out, sink);
                } else if (variant == 1) {
                    hipLaunchKernelGGL((fused_kernel<UA, XA, UB, XB>), dim3(NB), dim3(512), 0, s, WA[l], xin, WB[l], mid, out, bar, sink);
                } else if (variant == 2) {
                    hipLaunchKernelGGL((op_kernel<UB, XB>), dim3(NB), dim3(512), 0, s, WB[l], (const v4i*)mid, out, sink);       // op B alone
                } else {
                    hipLaunchKernelGGL((fused_ctrl_kernel<UA, XA, UB, XB>), dim3(NB), dim3(576), 0, s, WA[l], xin, WB[l], mid, out, bar, sink);
                }
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(gx, s));
            CK(hipStreamSynchronize(s));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int R = 50;
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(gx, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned err = 0; CK(hipMemcpy(&err, &bar->err[0], 4, hipMemcpyDeviceToHost));
            printf("%-54s %7.2f us per pair%s\n", variant == 0 ? "two launches (A, then B)" : variant == 1 ? "one launch (B requested, A, barrier, B)" : variant == 2 ? "op B alone (one launch)" : "one launch, control wave (A, B requested, barrier, B)",
                   ms * 1e3 / R / L, err ? "  [BARRIER TIMEOUT]" : "");
            CK(hipGraphExecDestroy(gx)); CK(hipGraphDestroy(g));
        }
    return 0;
}
