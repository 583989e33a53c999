// Experiment (DESIGN 7 item 8 ii): the 256x256x64 bf16 GEMM tile with FOUR waves of 128x128 (one wave per SIMD, 256 accumulator registers per lane -
// AGPRs - and four fragment sets in VGPRs) against the shipped eight waves of 128x64 (two per SIMD, the guide's 8-phase template).  Per K tile a wave
// reads 32 fragments for 128 MFMAs (the 8-wave kernel: 24 for 64), i.e. a third less LDS traffic per FLOP and half the waves at every barrier; with
// nobody else on its SIMD the wave has to hide its own fragment latency: the reads of phase p + 1 are issued before the MFMAs of phase p
// (quadrant order Q00, Q01, Q10, Q11; next tile's A0 / B0 fragments during Q11).  Same LDS ring (two K tiles x four 16 KiB half-tiles, LDS-DMA,
// counted vmcnt, raw barriers), same swizzle, same MFMA (16x16x32, weights as the first operand), plain per-lane 8-byte stores as the epilogue:
// this measures the K loop.  C = A [M][K] . W [N][K]^T, bf16.
//   hipcc --offload-arch=gfx950 -O3 -o build_tools/gemm_4wave tools/gemm_4wave.hip && build_tools/gemm_4wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define HT (128 * 128)            // one half-tile: 128 rows x 128 B
#define TILE (4 * HT)             // A0 A1 B0 B1
#define BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256) void gemm4w_kernel(const bf16_t* A, const bf16_t* W, bf16_t* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, fr = lane & 15, fg = lane >> 4;
    const int tilesM = (M + 255) / 256, tilesN = (N + 255) / 256, nt = tilesM * tilesN;
    int id;
    { const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, loc = bid >> 3; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
    int tm, tn;
    { const int GM = 8, gsz = GM * tilesN, g = id / gsz, first = g * GM; const int gm = min(GM, tilesM - first), in = id - g * gsz; tm = first + in % gm; tn = in / gm; }
    const int m0 = tm * 256, n0 = tn * 256;
    // DMA: a half-tile is 16 pieces of 8 rows; this wave moves pieces 4 wid .. 4 wid + 3
    const int lr8 = lane >> 3, lc = (lane & 7) ^ lr8;
    unsigned offA[2][4], offB[2][4];                    // element offsets (32 bits: sixteen 64-bit pointers cost the loop a spill)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lrow = (wid * 4 + i) * 8 + lr8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (lrow >> 6) * 128 + h * 64 + (lrow & 63); m = m < M ? m : M - 1;
            int n = n0 + (lrow >> 6) * 128 + h * 64 + (lrow & 63); n = n < N ? n : N - 1;
            offA[h][i] = (unsigned)m * (unsigned)K + lc * 8; offB[h][i] = (unsigned)n * (unsigned)K + lc * 8;
        }
    }
    auto dma = [&](const bf16_t* base, const unsigned (&off)[4], int k0, int buf, int slot) {
        char* dst = smem + buf * TILE + slot * HT + wid * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[i] + k0), (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 afA[4][2], afB[4][2], b0[4][2], b1[4][2];
    auto rd = [&](int buf, int slot, int wq, bf16x8 (&f)[4][2]) {     // wq: wave row (A) / wave column (B)
        const char* s = smem + buf * TILE + slot * HT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wq * 64 + i * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) f[i][kk] = *(const bf16x8*)(s + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
        }
    };
    auto quad = [&](int mh, int nh, const bf16x8 (&af)[4][2], const bf16x8 (&b)[4][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[nh * 4 + ni][mh * 4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni][kk], af[mi][kk], acc[nh * 4 + ni][mh * 4 + mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
    const int nk = K / 64;
    auto koff = [&](int t) { return (t < nk ? t : nk - 1) * 64; };   // (past the end: the last tile again, into slots nobody reads - constant wait counts)
    // prologue: tiles 0 and 1 in consumption order A0 B0 | B1 | A1
#pragma unroll
    for (int t = 0; t < 2; ++t) { dma(A, offA[0], koff(t), t, 0); dma(W, offB[0], koff(t), t, 2); dma(W, offB[1], koff(t), t, 3); dma(A, offA[1], koff(t), t, 1); }
    wait_vm<24>();                         // A0, B0 of tile 0
    BARRIER();
    rd(0, 0, wr, afA); rd(0, 2, wc, b0); lgkm0();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1, kn = koff(kt + 2);
        // ph0: Q00 while B1's fragments arrive; A0 / B0 of this tile were read by every wave before this barrier: refill them
        wait_vm<20>();                     // B1(kt)
        BARRIER();
        dma(A, offA[0], kn, buf, 0); dma(W, offB[0], kn, buf, 2);
        rd(buf, 3, wc, b1);
        __builtin_amdgcn_sched_barrier(0);
        quad(0, 0, afA, b0);
        lgkm0();
        // ph1: Q01 while A1's fragments arrive
        wait_vm<24>();                     // A1(kt)
        BARRIER();
        dma(W, offB[1], kn, buf, 3);
        rd(buf, 1, wr, afB);
        __builtin_amdgcn_sched_barrier(0);
        quad(0, 1, afA, b1);
        lgkm0();
        // ph2: Q10 (last use of b0)
        BARRIER();
        dma(A, offA[1], kn, buf, 1);
        quad(1, 0, afB, b0);
        // ph3: Q11 while the next tile's A0 / B0 fragments arrive
        wait_vm<24>();                     // A0, B0 (kt + 1)
        BARRIER();
        rd(buf ^ 1, 0, wr, afA); rd(buf ^ 1, 2, wc, b0);
        __builtin_amdgcn_sched_barrier(0);
        quad(1, 1, afB, b1);
        lgkm0();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // epilogue: acc[nb][mb][j] = D[n = n0 + wc*128 + nb*16 + fg*4 + j][m = m0 + wr*128 + mb*16 + fr]
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = m0 + wr * 128 + mb * 16 + fr, n = n0 + wc * 128 + nb * 16 + fg * 4;
            if (m < M && n + 3 < N) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16_t)acc[nb][mb][j];
                *(bf16x4*)(C + (long)m * N + n) = o;
            }
        }
}

int main(int argc, char** argv) {
    struct Shape { int M, N, K; } shapes[] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {48000, 5120, 1280}, {48000, 1280, 5120}, {48000, 3840, 1280}};
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE));
    for (auto sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = (bf16_t)rnd();
        for (auto& v : hW) v = (bf16_t)rnd();
        bf16_t *dA, *dW, *dC;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        const int nt = ((M + 255) / 256) * ((N + 255) / 256);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w_kernel, dim3(nt), dim3(256), 2 * TILE, 0, dA, dW, dC, M, N, K);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int R = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(gemm4w_kernel, dim3(nt), dim3(256), 2 * TILE, 0, dA, dW, dC, M, N, K);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // spot check against fp64 on 64 outputs
        std::vector<bf16_t> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 64; ++t) {
            const int m = (t * 977 + 13) % M, n = (t * 613 + 7) % N;
            double ref = 0; for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
            const double got = (double)(float)hC[(size_t)m * N + n];
            const double err = fabs(got - ref) / (fabs(ref) + 1.0);
            if (err > worst) worst = err;
        }
        printf("four waves of 128x128: M %6d N %5d K %5d  %8.1f us  %7.1f TFLOP/s  (worst relative error of 64 spot checks %.2e)\n", M, N, K, ms * 1e3 / R,
               2.0 * M * N * K / (ms / R * 1e-3) / 1e12, worst);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC));
    }
    return 0;
}
