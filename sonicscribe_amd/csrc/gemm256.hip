// gemm256_kernel: 256x256x64 block tile, 8 waves (2 along M x 4 along N), each wave a 128x64 output as 8x4
// v_mfma_f32_16x16x32_bf16 accumulators; the large-shape path of launch_gemm (encoder / projector / prefill linears).
//
// Pipeline (one K tile = 4 phases, one raw s_barrier per phase, no vmcnt(0) in the main loop):
//   * the block tile of each operand is cut into two "half-tiles" of 128 rows (16 KiB): A0/A1 hold the first/second 64 rows
//     of each wave-row's 128 rows, B0/B1 the first/second 32 rows of each wave-column's 64 rows, so phase p of every wave
//     computes one 64x32 quadrant of its output from one A half and one B half:
//         ph0: read A0,B0 frags -> quadrant (A0,B0)     ph1: read B1 -> (A0,B1)     ph2: read A1 -> (A1,B1)     ph3: (A1,B0)
//     24 ds_read_b128 + 64 MFMA per wave per K tile (fragments are reused from registers across phases);
//   * LDS holds two K tiles x four half-tile slots (128 KiB).  A slot is re-filled for K tile t+2 right after the barrier that
//     follows its last read in K tile t (ph1: A0,B0; ph2: B1; ph3: A1), through global_load_lds_dwordx4 (2 per wave per
//     half-tile), so up to two K tiles of DMA are in flight and every load has >= 1.5 K tiles of MFMA time to land;
//   * a phase waits only for the half-tile(s) it is about to read with a COUNTED s_waitcnt vmcnt(N) (12 / 10 / 12 in steady
//     state: the number of younger DMA instructions this wave has issued), then the barrier makes the other waves' pieces
//     visible (LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt + a barrier the reader has passed).
// Swizzle, MFMA operand roles and epilogues are those of gemm_kernel (gemm.hip).
#include <type_traits>

#include "common.h"

#define T256 256
#define TBK 64
#define HT_BYTES (128 * TBK * 2)            // one half-tile: 128 rows x 128 B
#define TILE_BYTES (4 * HT_BYTES)           // A0 A1 B0 B1
#define SLOT_A0 0
#define SLOT_A1 1
#define SLOT_B0 2
#define SLOT_B1 3

// raw barrier + compiler-only memory fence: LDS-DMA issues and ds_reads must not be moved across it by hipcc
#define BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x TILE_BYTES, the only LDS object of the kernel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int tilesM = (a.M + T256 - 1) / T256, tilesN = (a.N + T256 - 1) / T256;
    const int nt = tilesM * tilesN;
    int id;
    {
        const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, loc = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int tm, tn;
    {
        const int GM = 8, gsz = GM * tilesN, g = id / gsz, first = g * GM;
        const int gm = min(GM, tilesM - first), in = id - g * gsz;
        tm = first + in % gm;
        tn = in / gm;
    }
    const int m0 = tm * T256, n0 = tn * T256;
    const bf16_t* A = a.A + (long)blockIdx.z * a.strideA;
    bf16_t* C = a.C + (long)blockIdx.z * a.strideC;
    const bf16_t* R = (EPI == EPI_BIAS_RESID) ? a.R + (long)blockIdx.z * a.strideR : nullptr;

    // ---- DMA sources: per half-tile this wave moves rows j*8 .. j*8+7 for j = 2*wid, 2*wid+1 of the 128-row image
    const int lr8 = lane >> 3, lc = (lane & 7) ^ lr8;
    const bf16_t* srcA[2][2];   // [half][i]
    const bf16_t* srcB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lrow = (wid * 2 + i) * 8 + lr8;                 // row in the half-tile image
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (lrow >> 6) * 128 + h * 64 + (lrow & 63); m = m < a.M ? m : a.M - 1;
            int n = n0 + (lrow >> 5) * 64 + h * 32 + (lrow & 31); n = n < a.N ? n : a.N - 1;
            srcA[h][i] = A + (long)m * a.lda + lc * 8;
            srcB[h][i] = a.W + (long)n * a.K + lc * 8;
        }
    }
    auto dma = [&](const bf16_t* const (&src)[2], int k0, int buf, int slot) {
        char* dst = smem + buf * TILE_BYTES + slot * HT_BYTES + wid * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };

    f32x4 acc[4][8];   // [n-block][m-block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool vtile = (EPI == EPI_QKV_VT) && (n0 >= a.n_split);
    bf16x8 af[4][2], b0[2][2], b1[2][2];
    auto read_a = [&](int buf, int half) {
        const char* s = smem + buf * TILE_BYTES + (half ? SLOT_A1 : SLOT_A0) * HT_BYTES;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wr * 64 + mi * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *(const bf16x8*)(s + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
        }
    };
    auto read_b = [&](int buf, int half, bf16x8 (&b)[2][2]) {
        const char* s = smem + buf * TILE_BYTES + (half ? SLOT_B1 : SLOT_B0) * HT_BYTES;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int row = wc * 32 + ni * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[ni][kk] = *(const bf16x8*)(s + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
        }
    };
    auto quad = [&](int mh, int nh, const bf16x8 (&b)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if (vtile) acc[nh * 2 + ni][mh * 4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi][kk], b[ni][kk], acc[nh * 2 + ni][mh * 4 + mi], 0, 0, 0);
                    else acc[nh * 2 + ni][mh * 4 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni][kk], af[mi][kk], acc[nh * 2 + ni][mh * 4 + mi], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    // one K tile.  W0/W1/W2: vmcnt counts of ph0/ph1/ph2; ISSUE: refill this buffer with K tile kt+2
    auto ktile = [&](int kt, auto w0, auto w1, auto w2, auto issue) {
        constexpr int W0 = decltype(w0)::value, W1 = decltype(w1)::value, W2 = decltype(w2)::value;
        constexpr bool ISSUE = decltype(issue)::value;
        const int buf = kt & 1, kn = (kt + 2) * TBK;
        // ph0
        wait_vm<W0>();
        BARRIER();
        read_a(buf, 0);
        read_b(buf, 0, b0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(0, 0, b0);
        // ph1
        wait_vm<W1>();
        BARRIER();
        if (ISSUE) { dma(srcA[0], kn, buf, SLOT_A0); dma(srcB[0], kn, buf, SLOT_B0); }
        read_b(buf, 1, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(0, 1, b1);
        // ph2
        wait_vm<W2>();
        BARRIER();
        if (ISSUE) dma(srcB[1], kn, buf, SLOT_B1);
        read_a(buf, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(1, 1, b1);
        // ph3
        BARRIER();
        if (ISSUE) dma(srcA[1], kn, buf, SLOT_A1);
        quad(1, 0, b0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I10 = std::integral_constant<int, 10>;
    using I12 = std::integral_constant<int, 12>;
    using Yes = std::true_type;
    using No = std::false_type;

    const int nk = a.K / TBK;   // >= 4 (launch_gemm)
    // prologue: K tiles 0 and 1, issue order = consumption order (A0,B0 | B1 | A1)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        dma(srcA[0], t * TBK, t, SLOT_A0); dma(srcB[0], t * TBK, t, SLOT_B0);
        dma(srcB[1], t * TBK, t, SLOT_B1);
        dma(srcA[1], t * TBK, t, SLOT_A1);
    }
    int kt = 0;
    for (; kt < nk - 2; ++kt) ktile(kt, I12{}, I10{}, I12{}, Yes{});
    ktile(kt, I12{}, I10{}, I8{}, No{});      // K tile nk-2: nothing younger than tile nk-1's 8 DMA instructions
    ++kt;
    ktile(kt, I4{}, I2{}, I0{}, No{});        // K tile nk-1

    // ---- epilogue (as gemm_kernel): acc[nb][mb][j] = D[n = n0 + wc*64 + nb*16 + fg*4 + j][m = m0 + wr*128 + mb*16 + fr]
    if (vtile) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int n = n0 + wc * 64 + nb * 16 + fr;
            const float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int m = m0 + wr * 128 + mb * 16 + fg * 4;
                if (m < a.M && n < a.N) {
                    const int seg = m / a.seg_T, t = m - seg * a.seg_T;
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[nb][mb][j] + bv);
                    *(bf16x4*)(a.Vt + (long)seg * a.vt_seg_stride + (long)(n - a.n_split) * a.vt_ld + t) = o;
                }
            }
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oc = ((n0 + wc * 64) >> 1) + q * 16 + fg * 4;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int m = m0 + wr * 128 + mb * 16 + fr;
                if (m < a.M && (n0 + wc * 64 + q * 32) < a.N) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = rbf(acc[2 * q][mb][j]), u = rbf(acc[2 * q + 1][mb][j]);
                        o[j] = f2bf(rbf(silu_f(g)) * u);
                    }
                    *(bf16x4*)(C + (long)m * a.ldc + oc) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int n = n0 + wc * 64 + nb * 16 + fg * 4;
        if (n >= a.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const f32x4 b4 = *(const f32x4*)(a.bias + n);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = m0 + wr * 128 + mb * 16 + fr;
            if (m >= a.M) continue;
            bf16x4 o;
            if (EPI == EPI_BIAS_RESID) {
                const bf16x4 rv = *(const bf16x4*)(R + (long)m * a.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(rbf(acc[nb][mb][j] + bv[j]) + bf2f(rv[j]));
            } else if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(gelu_erf(rbf(acc[nb][mb][j] + bv[j])));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[nb][mb][j] + bv[j]);
            }
            *(bf16x4*)(C + (long)m * a.ldc + n) = o;
        }
    }
}

template <int EPI> static void launch256(const GemmArgs& a, hipStream_t s) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE_BYTES); attr = true; }
    const int tilesM = (a.M + T256 - 1) / T256, tilesN = (a.N + T256 - 1) / T256;
    hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(tilesM * tilesN, 1, a.batch > 0 ? a.batch : 1), dim3(512), 2 * TILE_BYTES, s, a);
}

bool gemm256_eligible(const GemmArgs& a, int epi) {
    if (a.K % TBK || a.K / TBK < 4) return false;
    if (a.M < 512 || a.N < 256) return false;
    if (epi == EPI_QKV_VT && (a.n_split % T256)) return false;
    return true;
}

void launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: launch256<EPI_BIAS>(a, s); break;
        case EPI_BIAS_GELU: launch256<EPI_BIAS_GELU>(a, s); break;
        case EPI_BIAS_RESID: launch256<EPI_BIAS_RESID>(a, s); break;
        case EPI_SWIGLU: launch256<EPI_SWIGLU>(a, s); break;
        case EPI_QKV_VT: launch256<EPI_QKV_VT>(a, s); break;
    }
}
