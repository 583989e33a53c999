// "one decoder layer as ONE persistent launch (loader + consumer waves, counter hand-offs) ... kept only if the step is <= 1.20 ms at
// 32 rows, otherwise the record closes the question with numbers."  This is the skeleton of that engine with everything a real one could be given for
// free, against the skeleton of the launch chain the engine runs, on the same bytes:
//   * the whole decode step (28 layers x 6 dependent ops: qkv, attention, o_proj, gate/up, down, add+norm) is ONE launch of 256 blocks (one per CU),
//     576 threads: eight working waves + a control wave that owns the grid barrier (it never has a load in flight, so its polls do not queue behind
//     prefetched weights - vmcnt retires in order);
//   * an op is pure reads: its share of the weights (or of the KV cache) by nontemporal 16-byte loads - REQUESTED BEFORE the barrier that precedes the op,
//     i.e. the prefetch a launch chain cannot do - then, behind the barrier, the activation image the previous op published (L2 / Infinity Cache), a fold
//     and 2 KiB of output per block.  No MFMA, no LDS staging, no softmax, no reduction trees: a real layer can only be slower;
//   * the grid barrier is the XCD-hierarchical one (8 groups of 32 blocks, one counter line each, one top counter, one generation word), agent-scope
//     release / acquire around it - what makes op N's stores visible to op N + 1's blocks on other XCDs.
// Variant 0 is the chain: the same ops as 6 x 28 kernels in one hipGraph (what tools/decode_floor.hip measures).  Byte counts: the full-size decoder at
// context 320, 32 or 64 rows (argv[1]).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/plf tools/persistent_layer_floor.hip && /tmp/plf 32 && /tmp/plf 64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
struct Bar { unsigned cnt[8][32]; unsigned top[32]; unsigned gen[32]; unsigned err[32]; };   // every word on a line of its own

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// loads per lane (16 B each, 512 lanes): W = the block's share of the op's weights / KV, X = the activation image it reads behind the barrier
//                     qkv  attn  o   gu  down add+norm
constexpr int WU[6] = {6,   11,   4,  24, 12,  0};          // 12.6 / 22 (KV at 32 rows) / 8.4 / 50.3 / 25.2 MB over 256 blocks
constexpr int XU32[6] = {4,  1,   8,  16, 6,   2};          // 32 rows: 32 / 6 / 64 / 128 / 48 KB images, 8 slabs of one row
constexpr int XU64[6] = {8,  1,   16, 32, 12,  2};

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

// ---- the chain: one kernel per op ----
template <int U, int XU> __global__ __launch_bounds__(512) void op_kernel(const v4i* w, const v4i* x, int* out, int* sink, int kvx) {
    v4i wv[U > 0 ? U : 1], xv[XU];
    issue<XU>(x + threadIdx.x, xv);
    v4i a = fold<XU>(xv);
    if constexpr (U > 0) {
        for (int r = 0; r < kvx; ++r) { issue<U>(w + ((long)blockIdx.x * kvx + r) * U * 512 + threadIdx.x, wv); a ^= fold<U>(wv); }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
    if (a[0] == 0x5a17c0de) sink[0] = 1;
}

// ---- the persistent step ----
__device__ __forceinline__ void grid_barrier_ctrl(Bar* b, unsigned& g, bool ctrl) {
    __builtin_amdgcn_s_barrier();                     // every working wave has drained its own stores (it waited vmcnt before requesting the next weights)
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

struct StepArgs { const v4i* w[6]; long wstride[6]; int* act[2]; Bar* bar; int* sink; int layers; int kvx; };

// op `O` of a layer: its weights are already in wv (requested before the barrier the caller has just passed); reads the image, publishes, drains its
// stores, and requests the NEXT op's weights into nv before returning
template <int O, bool R64> struct OpU { static constexpr int W = WU[O], X = R64 ? XU64[O] : XU32[O]; };

template <bool R64>
__global__ __launch_bounds__(576) void step_kernel(StepArgs a) {
    const bool ctrl = threadIdx.x >= 512;
    unsigned g = 0;
    if (ctrl) g = ld_sc1(&a.bar->gen[0]);
    const int t = threadIdx.x & 511;
    v4i w0[WU[0]], w1[WU[1]], w2[WU[2]], w3[WU[3]], w4[WU[4]];
    if (!ctrl) issue<WU[0]>(a.w[0] + (long)blockIdx.x * WU[0] * 512 + t, w0);
    int cur = 0;
    for (int l = 0; l < a.layers; ++l) {
        const long lo = (long)l;
#define STAGE(O, WV, NEXT_ISSUE) do { \
        grid_barrier_ctrl(a.bar, g, ctrl); \
        if (!ctrl) { \
            constexpr int XN = OpU<O, R64>::X; \
            v4i xv[XN]; \
            _Pragma("unroll") for (int u = 0; u < XN; ++u) xv[u] = ((const v4i*)a.act[cur])[u * 512 + t];   /* plain loads behind the acquire */ \
            v4i acc = fold<XN>(xv); \
            WV; \
            a.act[cur ^ 1][blockIdx.x * 512 + t] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3]; \
            if (acc[0] == 0x5a17c0de) a.sink[0] = 1; \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* this wave's stores have left before it requests anything new (in-order vmcnt) */ \
            NEXT_ISSUE; \
        } \
        cur ^= 1; } while (0)
        STAGE(0, acc ^= fold<WU[0]>(w0), issue<WU[1]>(a.w[1] + lo * a.wstride[1] + (long)blockIdx.x * a.kvx * WU[1] * 512 + t, w1));
        // attention: at 64 rows a block reads two rows' KV (kvx = 2): the second share is requested when the first has been folded
        STAGE(1, { acc ^= fold<WU[1]>(w1); for (int r = 1; r < a.kvx; ++r) { issue<WU[1]>(a.w[1] + lo * a.wstride[1] + ((long)blockIdx.x * a.kvx + r) * WU[1] * 512 + t, w1); acc ^= fold<WU[1]>(w1); } },
              issue<WU[2]>(a.w[2] + lo * a.wstride[2] + (long)blockIdx.x * WU[2] * 512 + t, w2));
        STAGE(2, acc ^= fold<WU[2]>(w2), issue<WU[3]>(a.w[3] + lo * a.wstride[3] + (long)blockIdx.x * WU[3] * 512 + t, w3));
        STAGE(3, acc ^= fold<WU[3]>(w3), issue<WU[4]>(a.w[4] + lo * a.wstride[4] + (long)blockIdx.x * WU[4] * 512 + t, w4));
        STAGE(4, acc ^= fold<WU[4]>(w4), (void)0);
        STAGE(5, (void)0, if (l + 1 < a.layers) issue<WU[0]>(a.w[0] + (lo + 1) * a.wstride[0] + (long)blockIdx.x * WU[0] * 512 + t, w0));
#undef STAGE
    }
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 32;
    const bool r64 = rows > 32;
    const int L = 28, NB = 256, kvx = r64 ? 2 : 1;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    v4i* W[6] = {nullptr}; long stride[6] = {0}; double mb = 0;
    for (int o = 0; o < 5; ++o) {
        stride[o] = (long)NB * WU[o] * 512 * (o == 1 ? kvx : 1);
        const size_t bytes = (size_t)stride[o] * 16 * L;
        CK(hipExtMallocWithFlags((void**)&W[o], bytes, hipDeviceMallocUncached)); CK(hipMemsetAsync(W[o], 0x11 + o, bytes, s));
        mb += bytes / 1e6;
    }
    int *act0, *act1, *sink; Bar* bar;
    CK(hipMalloc(&act0, 1 << 20)); CK(hipMalloc(&act1, 1 << 20)); CK(hipMemsetAsync(act0, 0x33, 1 << 20, s)); CK(hipMemsetAsync(act1, 0x44, 1 << 20, s));
    CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMemsetAsync(bar, 0, sizeof(Bar), s));
    CK(hipStreamSynchronize(s));
    printf("%d rows: %.0f MB of weights + KV per step, %d layers x 6 ops\n", rows, mb, L);
    for (int round = 0; round < 2; ++round)
        for (int variant = 0; variant < 2; ++variant) {
            hipGraph_t g; hipGraphExec_t gx;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            if (variant == 0) {
                int* act[2] = {act0, act1}; int cur = 0;
                for (int l = 0; l < L; ++l) {
#define OP(O) do { if (r64) hipLaunchKernelGGL((op_kernel<WU[O], XU64[O]>), dim3(NB), dim3(512), 0, s, W[O] ? W[O] + l * stride[O] : nullptr, (const v4i*)act[cur], act[cur ^ 1], sink, O == 1 ? kvx : 1); \
                     else hipLaunchKernelGGL((op_kernel<WU[O], XU32[O]>), dim3(NB), dim3(512), 0, s, W[O] ? W[O] + l * stride[O] : nullptr, (const v4i*)act[cur], act[cur ^ 1], sink, 1); cur ^= 1; } while (0)
                    OP(0); OP(1); OP(2); OP(3); OP(4); OP(5);
#undef OP
                }
            } else {
                StepArgs a{}; for (int o = 0; o < 6; ++o) { a.w[o] = W[o]; a.wstride[o] = stride[o]; }
                a.act[0] = act0; a.act[1] = act1; a.bar = bar; a.sink = sink; a.layers = L; a.kvx = kvx;
                if (r64) hipLaunchKernelGGL(step_kernel<true>, dim3(NB), dim3(576), 0, s, a); else hipLaunchKernelGGL(step_kernel<false>, dim3(NB), dim3(576), 0, s, a);
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(gx, s));
            CK(hipStreamSynchronize(s));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int R = 30;
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(gx, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned err = 0; CK(hipMemcpy(&err, &bar->err[0], 4, hipMemcpyDeviceToHost));
            printf("%-72s %8.1f us per step  (%.2f us per layer, %.2f TB/s)%s\n",
                   variant == 0 ? "chain: 6 launches per layer in one hipGraph (pure reads)" : "ONE persistent launch: weights prefetched across 6 grid barriers per layer",
                   ms * 1e3 / R, ms * 1e3 / R / L, mb / (ms / R) / 1e3, err ? "  [BARRIER TIMEOUT]" : "");
            CK(hipGraphExecDestroy(gx)); CK(hipGraphDestroy(g));
        }
    return 0;
}
