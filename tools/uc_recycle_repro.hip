// Attempt to isolate the corruption that made engine.cpp park uncached (hipDeviceMallocUncached) blocks in a process-lifetime pool
// instead of hipFree-ing them: test buffers that reused formerly-uncached pages read back wrong cache lines (round 2).
// Hypothesis: an XCD's L2 keeps lines of a physical page from an earlier ORDINARY tenant; the page is then mapped uncached and written
// (stores bypass L2, the old lines stay); freed; mapped ordinary again and filled by a host->device copy; a kernel then reads the stale
// L2 lines.  Each round:  ordinary alloc A -> kernel reads + writes A (lines resident in L2) -> free -> uncached alloc U of the same size
// -> kernel writes U -> free -> ordinary alloc B of the same size -> hipMemcpyAsync H2D pattern -> kernel counts words that differ.
// Variants: with / without the uncached tenant in between, with / without a system-scope cache flush kernel before the last free.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ucr tools/uc_recycle_repro.hip && /tmp/ucr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void touch(unsigned* p, size_t n, unsigned v) {          // read-modify-write: lines become resident (and dirty) in L2
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 3u + v;
}
__global__ void fill(unsigned* p, size_t n, unsigned v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}
__global__ void fill7(unsigned* p, size_t n, unsigned v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i * 7u;
}
__global__ void check(const unsigned* p, size_t n, unsigned v, unsigned long long* bad, unsigned long long* first) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (p[i] != v + (unsigned)i * 7u) { if (atomicAdd(bad, 1ull) == 0) *first = i; }
}
__global__ void flush_all() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, ""); }     // system scope: write back + invalidate, on every XCD

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *bad, *first; CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 8));
    const size_t sizes[] = {8u << 20, 24u << 20, 96u << 20, 3u << 20};
    std::vector<unsigned> host;
    for (int variant = 0; variant < 3; ++variant) {                  // 0: uncached tenant in between; 1: no uncached tenant (control); 2: as 0 + flush
        unsigned long long total_bad = 0; int same_addr = 0, rounds = 0;
        for (int it = 0; it < 24; ++it) {
            const size_t bytes = sizes[it % 4], n = bytes / 4;
            unsigned *A, *U = nullptr, *B;
            CK(hipMalloc(&A, bytes));
            hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, A, n, 17u + it);
            hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, A, n, 5u);
            CK(hipStreamSynchronize(s));
            void* a_addr = A;
            CK(hipFree(A));
            if (variant != 1) {
                CK(hipExtMallocWithFlags((void**)&U, bytes, hipDeviceMallocUncached));
                hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, U, n, 0xabcd0000u + it);
                CK(hipStreamSynchronize(s));
                if (variant == 2) { hipLaunchKernelGGL(flush_all, dim3(2048), dim3(64), 0, s); CK(hipStreamSynchronize(s)); }
                CK(hipFree(U));
            }
            CK(hipMalloc(&B, bytes));
            same_addr += ((void*)B == a_addr) || ((void*)B == (void*)U);
            host.resize(n);
            const unsigned v = 0x51000000u + it;
            for (size_t i = 0; i < n; ++i) host[i] = v + (unsigned)i * 7u;
            CK(hipMemcpyAsync(B, host.data(), bytes, hipMemcpyHostToDevice, s));
            CK(hipMemsetAsync(bad, 0, 8, s));
            hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, s, B, n, v, bad, first);
            unsigned long long hb = 0, hf = 0;
            CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(&hf, first, 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (hb) printf("  variant %d round %d (%zu MiB): %llu wrong words, first at word %llu\n", variant, it, bytes >> 20, hb, hf);
            total_bad += hb; ++rounds;
            CK(hipFree(B));
        }
        printf("%s: %llu wrong words over %d rounds (%d reused an address)\n",
               variant == 0 ? "uncached tenant between two ordinary ones" : variant == 1 ? "control: no uncached tenant" : "uncached tenant + system-scope flush before its free",
               total_bad, rounds, same_addr);
    }
    // Hypothesis 2: the ordinary tenant leaves DIRTY lines in an XCD's L2 (write-back cache, nothing flushes it when the memory is
    // freed); the uncached tenant writes memory directly; when the old dirty lines are evicted later they overwrite the uncached
    // tenant's data.  ordinary A: kernel writes (dirty) -> free -> uncached U at the same address: fill -> thrash the caches with a
    // 1 GiB ordinary buffer -> verify U.
    {
        unsigned* big; const size_t bigb = (size_t)1 << 30; CK(hipMalloc(&big, bigb));
        unsigned long long total_bad = 0; int same = 0;
        for (int it = 0; it < 40; ++it) {
            const size_t bytes = sizes[it % 4], n = bytes / 4;
            unsigned *A, *U;
            CK(hipMalloc(&A, bytes));
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, A, n, 0x77000000u + it);       // dirty lines of A in L2
            CK(hipStreamSynchronize(s));
            void* a_addr = A;
            CK(hipFree(A));
            CK(hipExtMallocWithFlags((void**)&U, bytes, hipDeviceMallocUncached));
            same += (void*)U == a_addr;
            const unsigned v = 0x51000000u + it;
            hipLaunchKernelGGL(fill7, dim3(1024), dim3(256), 0, s, U, n, v);
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, s, big, bigb / 4, 3u);             // evict whatever the caches still hold
            CK(hipMemsetAsync(bad, 0, 8, s));
            hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, s, U, n, v, bad, first);
            unsigned long long hb = 0, hf = 0;
            CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(&hf, first, 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (hb) printf("  dirty-line round %d (%zu MiB): %llu wrong words in the uncached tenant, first at word %llu\n", it, bytes >> 20, hb, hf);
            total_bad += hb;
            CK(hipFree(U));
        }
        printf("uncached tenant behind a dirty ordinary one, caches thrashed: %llu wrong words over 40 rounds (%d reused the address)\n", total_bad, same);
        CK(hipFree(big));
    }
    return 0;
}
