// Can the decode step's kernel boundaries be taken off the critical path WITHOUT a persistent kernel?
// Idea: launch the step's kernels with hipExtAnyOrderLaunch (AQL packets without the barrier bit: the command processor starts
// dispatching kernel i+1 as soon as every block of kernel i has been DISPATCHED, not completed) and order them by hand: the blocks of
// kernel i+1 request their weights first - the part of the work that depends on nothing - and only then wait for a counter that the blocks
// of kernel i bump after their last store (release / acquire at agent scope).  Launch latency, the kernel's head and most of its HBM
// stream then overlap the previous kernel; what stays serial per edge is release + counter + acquire + the small activation read.
// In-order dispatch inside one hardware queue makes the spin-wait deadlock-free (every block of the producer is resident or ahead in
// the dispatch order); every spin is bounded by the wall clock anyway and reports a timeout.
//
// Same byte counts and launch count as tools/decode_floor.hip (28 layers x [qkv 12.6 MB, KV 22 MB on 128 blocks, o 8.4 MB, gate/up 50 MB,
// down 25 MB, tiny] + lm_head 243 MB + tiny).  Variants:
//   graph      ordinary launches captured in a hipGraph (the engine's form; decode_floor's number)
//   direct     ordinary launches from the host, no graph
//   anyorder   any-order launches + counters, weights requested BEFORE the wait
//   anyorder-late   the same with the weights requested AFTER the wait (isolates what the early request buys)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/aoc tools/anyorder_chain.hip && timeout 120 /tmp/aoc
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

struct Link {
    const v4i* w; long n16;            // this kernel's weights
    const v4i* x_in; v4i* x_out;       // small activation buffers (x_in written by the previous kernel)
    int* prev; int expect;             // wait until *prev >= expect (prev == null: no wait)
    int* mine;                         // bumped once per block at the end
    int* err;                          // err[0] += 1 on a spin timeout
    int early;                         // weights requested before (1) / after (0) the wait
    int fences;                        // 1: agent-scope release / acquire fences (ordinary memory); 0: none (x buffers and counters are UNCACHED
                                       // allocations: every access goes to memory, vmcnt(0) before the counter bump is the whole release)
};

template <int U>
__global__ __launch_bounds__(512) void link_kernel(Link a) {
    extern __shared__ char lds[];
    const long per = (a.n16 + gridDim.x - 1) / gridDim.x;
    const long lo = per * blockIdx.x, hi = lo + per < a.n16 ? lo + per : a.n16;
    v4i acc = {0, 0, 0, 0};
    v4i v[U];
    const int T = blockDim.x;
    auto request = [&](long base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = base + (long)u * T;
            v[u] = j < hi ? __builtin_nontemporal_load(a.w + j) : (v4i){0, 0, 0, 0};
        }
    };
    if (a.early) request(lo + threadIdx.x);
    if (a.prev) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(a.prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.expect && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > 20000000ll) { atomicAdd(a.err, 1); break; }     // 0.2 s at 100 MHz
            }
            if (a.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE acquire per block (invalidates the CU's vector cache and the XCD's L2 lines)
        }
        __syncthreads();
    }
    if (!a.early) request(lo + threadIdx.x);
    // the activation slice: 64 bytes per thread (32 KiB per block), written by the previous kernel
    v4i x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = a.x_in[(threadIdx.x + i * 512) & 8191];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= x[i];
    for (long i = lo + threadIdx.x + (long)U * T; i < hi; i += (long)U * T) {        // (shares larger than U pieces: lm_head)
        request(i);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if (threadIdx.x < 32) a.x_out[(blockIdx.x * 32 + threadIdx.x) & 8191] = acc;
    __syncthreads();                                   // every wave's stores have been acknowledged by L2 (vmcnt(0) + barrier)
    if (threadIdx.x == 0 && !a.fences && a.mine) __hip_atomic_fetch_add(a.mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && a.fences && a.mine) __hip_atomic_fetch_add(a.mine, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);    // ONE release per block: an agent-scope
                                                       // fence writes the XCD's L2 back; executed by every wave it costs ~35 us per kernel
}

struct K { size_t bytes; int blocks, lds; };

template <int U> static void launch(const Link& a, const K& k, hipStream_t s, bool anyorder) {
    if (k.lds > 65536) {
        static bool done = false;
        if (!done) { CK(hipFuncSetAttribute((const void*)link_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000)); done = true; }
    }
    if (anyorder) hipExtLaunchKernelGGL((link_kernel<U>), dim3(k.blocks), dim3(512), k.lds, s, nullptr, nullptr, hipExtAnyOrderLaunch, a);
    else hipLaunchKernelGGL((link_kernel<U>), dim3(k.blocks), dim3(512), k.lds, s, a);
}
static void launch_u(const Link& a, const K& k, hipStream_t s, bool anyorder) {
    const size_t per_thread = k.bytes / k.blocks / 512 / 16;
    if (per_thread <= 2) launch<2>(a, k, s, anyorder);
    else if (per_thread <= 6) launch<6>(a, k, s, anyorder);
    else if (per_thread <= 12) launch<12>(a, k, s, anyorder);
    else launch<24>(a, k, s, anyorder);
}

int main() {
    const int L = 28, B = 32, ctx = 335;
    const size_t qkv = 3072ul * 2048 * 2, wo = 2048ul * 2048 * 2, gu = 2ul * 6144 * 2048 * 2, dn = 2048ul * 6144 * 2, head = 59264ul * 2048 * 2;
    const size_t kv = (size_t)B * 4 * ctx * 128 * 2 * 2;
    // LDS footprints of the real kernels decide which neighbours can be co-resident on a CU (160 KiB)
    const K ks[6] = {{qkv, 256, 32768}, {kv, 128, 20480}, {wo, 256, 65536}, {gu, 256, 131072}, {dn, 256, 49152}, {32768, 32, 1024}};
    const K khead = {head, 256, 65536}, ktiny = {32768, 32, 1024};
    const size_t per_layer = qkv + kv + wo + gu + dn + 32768;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<char*> W(L); char* H;
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&W[l], per_layer)); CK(hipMemsetAsync(W[l], 0x11, per_layer, s)); }
    CK(hipMalloc(&H, head)); CK(hipMemsetAsync(H, 0x11, head, s));
    v4i *xa, *xb, *xa_uc, *xb_uc, *xa_c, *xb_c;
    CK(hipMalloc(&xa_c, 8192 * 16)); CK(hipMalloc(&xb_c, 8192 * 16));
    CK(hipExtMallocWithFlags((void**)&xa_uc, 8192 * 16, hipDeviceMallocUncached)); CK(hipExtMallocWithFlags((void**)&xb_uc, 8192 * 16, hipDeviceMallocUncached));
    xa = xa_c; xb = xb_c;
    CK(hipMemsetAsync(xa_c, 0, 8192 * 16, s)); CK(hipMemsetAsync(xb_c, 0, 8192 * 16, s)); CK(hipMemsetAsync(xa_uc, 0, 8192 * 16, s)); CK(hipMemsetAsync(xb_uc, 0, 8192 * 16, s));
    const int NK = L * 6 + 2;
    int *ctr, *ctr_c, *ctr_uc; CK(hipMalloc(&ctr_c, (NK + 1) * 64 * 4)); CK(hipExtMallocWithFlags((void**)&ctr_uc, (NK + 1) * 64 * 4, hipDeviceMallocUncached)); ctr = ctr_c;
    int* err; CK(hipMalloc(&err, 64));
    int fences = 1; bool bump = true;
    CK(hipStreamSynchronize(s));

    auto enqueue_step = [&](int step, bool anyorder, bool wait, int early) {
        int idx = 0;
        auto one = [&](const char* w, const K& k) {
            Link a{};
            a.w = (const v4i*)w; a.n16 = (long)(k.bytes / 16); a.x_in = (idx & 1) ? xb : xa; a.x_out = (idx & 1) ? xa : xb;
            auto blocks_of = [&](int j) { return j < L * 6 ? ks[j % 6].blocks : j == L * 6 ? khead.blocks : ktiny.blocks; };
            const int prev_idx = idx == 0 ? NK - 1 : idx - 1;
            a.prev = wait ? ctr + prev_idx * 64 : nullptr;
            // counters are monotonic: after `step` full steps kernel j has been bumped blocks_j * step times; the first kernel of a step waits
            // for the last kernel of the previous step
            a.expect = blocks_of(prev_idx) * (idx == 0 ? step : step + 1);
            a.mine = bump ? ctr + idx * 64 : nullptr; a.err = err; a.early = early; a.fences = fences;
            launch_u(a, k, s, anyorder);
            ++idx;
        };
        for (int l = 0; l < L; ++l) {
            const char* p = W[l];
            for (int k = 0; k < 6; ++k) { one(p, ks[k]); p += ks[k].bytes; }
        }
        one(H, khead);
        one(W[0], ktiny);
    };
    auto reset = [&]() { CK(hipMemsetAsync(ctr_c, 0, (NK + 1) * 64 * 4, s)); CK(hipMemsetAsync(ctr_uc, 0, (NK + 1) * 64 * 4, s)); CK(hipMemsetAsync(err, 0, 64, s)); CK(hipStreamSynchronize(s)); };
    const double bytes = (double)per_layer * L + head;
    auto report = [&](const char* name, double us, double host_us) {
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%-16s %8.1f us per step = %.2f TB/s  (host enqueue %.1f us per step, spin timeouts %d)\n", name, us, bytes / us / 1e6, host_us, herr); fflush(stdout);
    };
    const int reps = 30;
    hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    for (int gb = 0; gb < 2; ++gb) {   // graph of ordinary launches (no waits: the stream orders them), without / with the counter bump + release
        bump = gb == 1;
        reset();
        hipGraph_t g; hipGraphExec_t gx;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue_step(0, false, false, 1);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(gx, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(ea, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(gx, s));
        CK(hipEventRecord(eb, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        report(gb ? "graph+release" : "graph", ms * 1e3 / reps, 0.0);
        CK(hipGraphExecDestroy(gx)); CK(hipGraphDestroy(g));
    }
    bump = true;
    for (int mem = 0; mem < 2; ++mem) {
    if (mem == 1) { xa = xa_uc; xb = xb_uc; ctr = ctr_uc; fences = 0; printf("-- activations and counters uncached, no fences --\n"); }
    for (int variant = 0; variant < 5; ++variant) {
        const bool anyorder = variant >= 1, wait = variant >= 1 && variant != 4; const int early = variant == 2 ? 0 : 1;
        // variant 3 = variant 1 again (run-to-run spread); variant 4: any-order launches WITHOUT the waits (a race, timing only: how much do
        // the launches overlap when nothing holds them back?)
        reset();
        int step = 0;
        for (int i = 0; i < 3; ++i) enqueue_step(step++, anyorder, wait, early);
        CK(hipStreamSynchronize(s));
        const auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(ea, s));
        for (int i = 0; i < reps; ++i) enqueue_step(step++, anyorder, wait, early);
        CK(hipEventRecord(eb, s));
        const auto h1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        report(variant == 0 ? "direct" : variant == 2 ? "anyorder-late" : variant == 4 ? "anyorder-nowait" : "anyorder", ms * 1e3 / reps, std::chrono::duration<double, std::micro>(h1 - h0).count() / reps);
    }
    }
    return 0;
}
