// gemm256_kernel: 256x256x64 block tile, 8 waves (2 along M x 4 along N), each wave a 128x64 output as 8x4
// v_mfma_f32_16x16x32_bf16 accumulators; the large-shape path of launch_gemm (encoder / projector / prefill linears).
//
// Pipeline (STAGGER = false: one K tile = 4 phases, one raw s_barrier per phase, no vmcnt(0) in the main loop):
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
// STAGGER = true (default): the two waves that share a SIMD (w and w+4) run half a phase apart -- waves 0-3 read a
// phase's fragments while waves 4-7 issue the previous phase's MFMAs and vice versa -- so LDS reads overlap the
// partner's matrix work instead of both waves of a SIMD reading, then both computing.  7 barriers per K tile:
//     slot   0      1      2      3      4      5      6
//     w0-3   R0     M0     R1     M1     R2     M2     M3          R0: A0,B0 frags  R1: B1  R2: A1
//     w4-7   M3'    R0     M0     R1     M1     R2     M2          Mi: 16 MFMA of quadrant i (M3' = previous K tile's)
//     DMA                  A0,B0         B1            A1           (refill for K tile t+2, after the last reader's barrier)
//     wait          D1            D2                   D0(t+1)      (counted vmcnt 10 / 12 / 12, identical for all waves)
// Swizzle, MFMA operand roles and epilogues are those of gemm_kernel (gemm.hip).
#include <type_traits>

#include "common.h"
#include "int8_util.h"

#define T256 256
#define TBKB 128                            // bytes of K per tile row: 64 16-bit or 128 int8 elements
#define HT_BYTES (128 * TBKB)                // one half-tile: 128 rows x 128 B
#define TILE_BYTES (4 * HT_BYTES)           // A0 A1 B0 B1
#define LDS256_BYTES (256 * 528)            // >= 2 * TILE_BYTES; also holds the staged output tile of the epilogue
#define SLOT_A0 0
#define SLOT_A1 1
#define SLOT_B0 2
#define SLOT_B1 3

// raw barrier + compiler-only memory fence: LDS-DMA issues and ds_reads must not be moved across it by hipcc
#define BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename KD, int EPI, bool STAGGER>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs a) {
    typedef typename KD::elem ET_; typedef typename KD::out OT; typedef typename KD::frag Frag; typedef typename KD::acc Acc;
    typedef typename ET<OT>::v4 O4; typedef typename ET<OT>::v8 O8;
    constexpr int EB = sizeof(ET_), CE = 16 / EB, TBK = TBKB / EB;   // bytes per element, elements per 16-B chunk / per K tile
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x TILE_BYTES, the only LDS object of the kernel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;
#define G256_STAMP(i) do { if (a.dbg && tid == 0) a.dbg[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    G256_STAMP(0);
    if (a.dbg && tid == 0) a.dbg[(long)blockIdx.x * 8 + 4] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID, HW_ID

    const int tilesM = (a.M + T256 - 1) / T256, tilesN = (a.N + T256 - 1) / T256;
    const int nt = tilesM * tilesN;
    int id;
    {
        const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, loc = bid >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int tm, tn;
    {
        const int GM = a.raster_gm > 0 ? a.raster_gm : 8, gsz = GM * tilesN, g = id / gsz, first = g * GM;
        const int gm = min(GM, tilesM - first), in = id - g * gsz;
        tm = first + in % gm;
        tn = in / gm;
    }
    const int m0 = tm * T256, n0 = tn * T256;
    const ET_* A = (const ET_*)a.A + (long)blockIdx.z * a.strideA;
    OT* C = (OT*)a.C + (long)blockIdx.z * a.strideC;
    const OT* R = (EPI == EPI_BIAS_RESID) ? (const OT*)a.R + (long)blockIdx.z * a.strideR : nullptr;

    // ---- DMA sources: per half-tile this wave moves rows j*8 .. j*8+7 for j = 2*wid, 2*wid+1 of the 128-row image
    const int lr8 = lane >> 3, lc = (lane & 7) ^ lr8;
    const ET_* srcA[2][2];   // [half][i]
    const ET_* srcB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lrow = (wid * 2 + i) * 8 + lr8;                 // row in the half-tile image
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int m = m0 + (lrow >> 6) * 128 + h * 64 + (lrow & 63); m = m < a.M ? m : a.M - 1;
            int n = n0 + (lrow >> 5) * 64 + h * 32 + (lrow & 31); n = n < a.N ? n : a.N - 1;
            srcA[h][i] = A + (long)m * a.lda + lc * CE;
            srcB[h][i] = (const ET_*)a.W + (long)n * a.K + lc * CE;
            if (a.w_tiled) {
                // fragment-tiled W: piece wid * 2 + i of the half-tile = (row tile wid of the image, k-step i): 1 KiB contiguous in memory, lane-linear
                int nb = n0 + (wid >> 1) * 64 + h * 32 + (wid & 1) * 16; nb = nb + 16 <= a.N ? nb : a.N - 16;
                srcB[h][i] = (const ET_*)a.W + ((long)(nb >> 4) * (a.K / (TBK / 2)) + i) * (TBK / 2 * 16) + lane * CE;
            }
        }
    }
    const int kmulB = a.w_tiled ? 16 : 1;        // elements the tiled source advances per element of k (a (16-row, k-step) tile is 16 x the k-step wide)
    auto dma = [&](const ET_* const (&src)[2], int k0, int buf, int slot) {
        char* dst = smem + buf * TILE_BYTES + slot * HT_BYTES + wid * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };

    // bf16 GELU table -> LDS behind the tile ring (9 KiB, one wave, issued first: older than every operand DMA, so the counted waits of
    // the K loop cover it)
    constexpr bool LUT = EPI == EPI_BIAS_GELU && std::is_same<KD, KBF16>::value;
    if constexpr (LUT) {
        if (wid == 0 && a.gelu_lut) {
#pragma unroll
            for (int i = 0; i < GELU_LUT_N * 2 / 1024; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)a.gelu_lut + i * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + LDS256_BYTES + i * 1024), 16, 0, 0);
        }
    }
    // The tile's 256 bias values -> LDS (16-bit kinds; 1 KiB = one DMA instruction of one wave, issued before the operand prologue: older than
    // every operand DMA, so the K loop's first counted wait + barrier cover it).  Round 4 loaded them with four dependent global loads per lane
    // AFTER the K loop: ~3 us of exposed load latency in every tile's epilogue.
    constexpr int BIAS_OFF = LDS256_BYTES + (LUT ? GELU_LUT_N * 2 : 0);
    constexpr bool SBIAS = !KD::I8;
    if constexpr (SBIAS) {
        if (wid == 1 && a.bias) {
            int n = n0 + lane * 4; n = n + 3 < a.N ? n : 0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias + n),
                                             (__attribute__((address_space(3))) void*)(smem + BIAS_OFF), 16, 0, 0);
        }
    }
    const float* sbias = (const float*)(smem + BIAS_OFF);
    Acc acc[4][8];   // [n-block][m-block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;

    // V tiles of the fused QKV GEMM leave transposed.  16-bit kinds compute them transposed (operands swapped in the MFMA); the int8 kind keeps
    // the normal orientation - its dequantising epilogue is written per row - and transposes while staging (vtile8)
    const bool vtile = (EPI == EPI_QKV_VT) && !KD::I8 && (n0 >= a.n_split);
    const bool vtile8 = (EPI == EPI_QKV_VT) && KD::I8 && (n0 >= a.n_split);
    Frag af[4][2], b0[2][2], b1[2][2];
    auto read_a = [&](int buf, int half) {
        const char* s = smem + buf * TILE_BYTES + (half ? SLOT_A1 : SLOT_A0) * HT_BYTES;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wr * 64 + mi * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[mi][kk] = *(const Frag*)(s + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
        }
    };
    // byte offsets of this lane's four B fragments in a half-tile: swizzled rows, or (tiled W) piece (row tile, k-step) + lane * 16
    int boff[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int row = wc * 32 + ni * 16 + fr;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) boff[ni][kk] = a.w_tiled ? ((wc * 2 + ni) * 2 + kk) * 1024 + lane * 16 : row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4);
    }
    auto read_b = [&](int buf, int half, Frag (&b)[2][2]) {
        const char* s = smem + buf * TILE_BYTES + (half ? SLOT_B1 : SLOT_B0) * HT_BYTES;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[ni][kk] = *(const Frag*)(s + boff[ni][kk]);
    };
    // vt (compile-time): transposed MFMA roles for the V tiles of the fused QKV GEMM; hoisted out of the K loop
    auto quad = [&](auto vt, int mh, int nh, const Frag (&b)[2][2]) {
        constexpr bool VT = decltype(vt)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if (VT) acc[nh * 2 + ni][mh * 4 + mi] = KD::mfma(af[mi][kk], b[ni][kk], acc[nh * 2 + ni][mh * 4 + mi]);
                    else acc[nh * 2 + ni][mh * 4 + mi] = KD::mfma(b[ni][kk], af[mi][kk], acc[nh * 2 + ni][mh * 4 + mi]);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    // one K tile.  W0/W1/W2: vmcnt counts of ph0/ph1/ph2; ISSUE: refill this buffer with K tile kt+2
    auto ktile = [&](auto vt, int kt, auto w0, auto w1, auto w2, auto issue) {
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
        quad(vt, 0, 0, b0);
        // ph1
        wait_vm<W1>();
        BARRIER();
        if (ISSUE) { dma(srcA[0], kn, buf, SLOT_A0); dma(srcB[0], kn * kmulB, buf, SLOT_B0); }
        read_b(buf, 1, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(vt, 0, 1, b1);
        // ph2
        wait_vm<W2>();
        BARRIER();
        if (ISSUE) dma(srcB[1], kn * kmulB, buf, SLOT_B1);
        read_a(buf, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(vt, 1, 1, b1);
        // ph3
        BARRIER();
        if (ISSUE) dma(srcA[1], kn, buf, SLOT_A1);
        quad(vt, 1, 0, b0);
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
        dma(srcA[0], t * TBK, t, SLOT_A0); dma(srcB[0], t * TBK * kmulB, t, SLOT_B0);
        dma(srcB[1], t * TBK * kmulB, t, SLOT_B1);
        dma(srcA[1], t * TBK, t, SLOT_A1);
    }
    if (!STAGGER) {
        auto run0 = [&](auto vt) {
            int kt = 0;
            for (; kt < nk - 2; ++kt) ktile(vt, kt, I12{}, I10{}, I12{}, Yes{});
            ktile(vt, kt, I12{}, I10{}, I8{}, No{});      // K tile nk-2: nothing younger than tile nk-1's 8 DMA instructions
            ++kt;
            ktile(vt, kt, I4{}, I2{}, I0{}, No{});        // K tile nk-1
        };
        if constexpr (EPI == EPI_QKV_VT) { if (vtile) run0(Yes{}); else run0(No{}); }
        else run0(No{});
    } else {
        auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
        // one K tile of one wave group (G0 = true: waves 0-3).  The two groups run separate straight-line loops (the branch
        // on the group is hoisted out of the K loop) with the same barrier / DMA / vmcnt sequence.
        // w1 / w2 / w0n: vmcnt before the barriers that open slots 2 / 4 / next tile's 0; first: no M3' yet
        auto stile = [&](auto grp, auto vt, int kt, auto w1, auto w2, auto w0n, auto issue, bool first) {
            constexpr bool G0 = decltype(grp)::value;
            constexpr int W1 = decltype(w1)::value, W2 = decltype(w2)::value, W0N = decltype(w0n)::value;
            constexpr bool ISSUE = decltype(issue)::value;
            const int buf = kt & 1, kn = (kt + 2) * TBK;
            BARRIER();                                                                   // slot 0
            if (G0) { read_a(buf, 0); read_b(buf, 0, b0); lgkm0(); } else if (!first) quad(vt, 1, 0, b0);
            BARRIER();                                                                   // slot 1
            if (G0) quad(vt, 0, 0, b0); else { read_a(buf, 0); read_b(buf, 0, b0); lgkm0(); }
            wait_vm<W1>();
            BARRIER();                                                                   // slot 2
            if (ISSUE) { dma(srcA[0], kn, buf, SLOT_A0); dma(srcB[0], kn * kmulB, buf, SLOT_B0); }
            if (G0) { read_b(buf, 1, b1); lgkm0(); } else quad(vt, 0, 0, b0);
            BARRIER();                                                                   // slot 3
            if (G0) quad(vt, 0, 1, b1); else { read_b(buf, 1, b1); lgkm0(); }
            wait_vm<W2>();
            BARRIER();                                                                   // slot 4
            if (ISSUE) dma(srcB[1], kn * kmulB, buf, SLOT_B1);
            if (G0) { read_a(buf, 1); lgkm0(); } else quad(vt, 0, 1, b1);
            BARRIER();                                                                   // slot 5
            if (G0) quad(vt, 1, 1, b1); else { read_a(buf, 1); lgkm0(); }
            BARRIER();                                                                   // slot 6
            if (ISSUE) dma(srcA[1], kn, buf, SLOT_A1);
            if (G0) quad(vt, 1, 0, b0); else quad(vt, 1, 1, b1);
            if (W0N >= 0) wait_vm<(W0N >= 0 ? W0N : 0)>();
        };
        using IM1 = std::integral_constant<int, -1>;
        auto run = [&](auto grp, auto vt) {
            wait_vm<12>();                                                               // D0(0) landed
            G256_STAMP(1);
            int kt = 0;
            for (; kt < nk - 2; ++kt) stile(grp, vt, kt, I10{}, I12{}, I12{}, Yes{}, kt == 0);
            stile(grp, vt, kt, I10{}, I8{}, I4{}, No{}, false);
            ++kt;
            stile(grp, vt, kt, I2{}, I0{}, IM1{}, No{}, false);
            if (!decltype(grp)::value) quad(vt, 1, 0, b0);                               // waves 4-7: last K tile's quadrant 3
        };
        if constexpr (EPI == EPI_QKV_VT) {
            if (vtile) { if (wid < 4) run(Yes{}, Yes{}); else run(No{}, Yes{}); }
            else { if (wid < 4) run(Yes{}, No{}); else run(No{}, No{}); }
        } else { if (wid < 4) run(Yes{}, No{}); else run(No{}, No{}); }
    }

    G256_STAMP(2);
    // ---- epilogue.  acc[nb][mb][j] = D[n = n0 + wc*64 + nb*16 + fg*4 + j][m = m0 + wr*128 + mb*16 + fr].
    // Per-lane stores of that layout are 8-byte pieces on 16 different lines per instruction (measured: ~18 us per tile,
    // a third of the kernel at K = 1280).  Instead the bf16 tile is staged in LDS (free once the K loop is done) with the
    // op's rounding sequence applied in registers, then written as full 512-byte rows, 16 B per lane; the residual is read
    // with the same coalesced pattern in that final pass.
    __syncthreads();
    // Residual pieces of the row-wise store pass (16-bit kinds): requested NOW, 16 x 16 bytes per thread, so that they arrive while the tile is
    // converted and staged.  Round 4 loaded each piece inside the store pass and waited for it there: 4-10 us of a residual tile's fixed cost
    // was exposed load latency.  asm loads + one hand-placed wait (the compiler would otherwise wait at its own first use - fine - but it may
    // also move plain loads below the staging code again).
    constexpr bool RPRE = EPI == EPI_BIAS_RESID && !KD::I8;
    i32x4 rpre[RPRE ? 16 : 1];
    if constexpr (RPRE) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int c = it * 512 + tid, row = c >> 5, ch = c & 31;
            int m = m0 + row, n = n0 + ch * 8;
            m = m < a.M ? m : a.M - 1; n = n < a.N ? n : 0;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rpre[it]) : "v"(R + (long)m * a.ldr + n) : "memory");
        }
    }
    constexpr int CLD = 528;                     // staged row pitch in bytes (256 bf16 + 16 B skew)
    if (EPI == EPI_QKV_VT && vtile) {
        // staged transposed: row = n (256), columns = m; V^T[seg][n - n_split][t .. t+3] leaves in 8-byte pieces (T % 4 == 0)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int nl = wc * 64 + nb * 16 + fr, n = n0 + nl;
            const float bv = (a.bias && n < a.N) ? (SBIAS ? sbias[nl] : a.bias[n]) : 0.f;
            const int nc = n < a.N ? n : a.N - 1;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                O4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int m = m0 + wr * 128 + mb * 16 + fg * 4 + j; m = m < a.M ? m : a.M - 1;
                    o[j] = (OT)gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[nb][mb][j], m, nc, bv, I8Row{0.f, 0, 0, false}, 0.f);   // (V^T tiles exist for 16-bit kinds only)
                }
                *(O4*)(smem + nl * CLD + (wr * 128 + mb * 16 + fg * 4) * 2) = o;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int c = it * 512 + tid, nl = c >> 6, mc = c & 63;
            const int m = m0 + mc * 4, n = n0 + nl;
            if (m < a.M && n < a.N) {
                const int seg = m / a.seg_T, t = m - seg * a.seg_T;
                *(O4*)((OT*)a.Vt + (long)seg * a.vt_seg_stride + (long)(n - a.n_split) * a.vt_ld + t) = *(const O4*)(smem + nl * CLD + mc * 8);
            }
        }
        return;
    }
    // int8: statistics of the 8 rows this lane touches, loaded once (not per element)
    I8Row rws[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) { int m = m0 + wr * 128 + mb * 16 + fr; m = m < a.M ? m : a.M - 1; rws[mb] = i8_row<KD>(a, m); }
    // deferred rows (their outlier columns and the residual are added by launch_i8_outlier_side): one flag byte per tile row behind the
    // staged tile, read back by the row-wise store pass
    constexpr bool DEFER = KD::I8 && EPI == EPI_BIAS_RESID;
    if constexpr (DEFER) {
        if (wc == 0 && fg == 0) {
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) smem[LDS256_BYTES + wr * 128 + mb * 16 + fr] = rws[mb].defer ? 1 : 0;
        }
    }
    if constexpr (KD::I8 && EPI == EPI_QKV_VT) {
        if (vtile8) {
            // int8 V tile: module outputs as in the bias epilogue (dequantisation + outlier columns of the row's request), staged TRANSPOSED
            // (row = n, 2-byte stores), then the same coalesced V^T rows out as the 16-bit kinds
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int nl = wc * 64 + nb * 16 + fg * 4, n = n0 + nl;
                const int nc = n + 3 < a.N ? n : 0;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) { const f32x4 b4 = *(const f32x4*)(a.bias + nc); bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3]; }
                const f32x4 sb = *(const f32x4*)(a.q.scb + nc);
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    int m = m0 + wr * 128 + mb * 16 + fr; m = m < a.M ? m : a.M - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(OT*)(smem + (nl + j) * CLD + (wr * 128 + mb * 16 + fr) * 2) = (OT)gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[nb][mb][j], m, nc + j, bv[j], rws[mb], sb[j]);
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < 32; ++it) {
                const int c = it * 512 + tid, nl = c >> 6, mc = c & 63;
                const int m = m0 + mc * 4, n = n0 + nl;
                if (m < a.M && n < a.N) {
                    const int seg = m / a.seg_T, t = m - seg * a.seg_T;
                    *(O4*)((OT*)a.Vt + (long)seg * a.vt_seg_stride + (long)(n - a.n_split) * a.vt_ld + t) = *(const O4*)(smem + nl * CLD + mc * 8);
                }
            }
            return;
        }
    }
    if (EPI == EPI_SWIGLU && a.gu8) {
        // gate / up in 8-row groups: accumulator block nb holds gate columns 8 nb' .. in lanes fg < 2 and their up partners in lanes fg >= 2
        // (rows 4 fg + j of the 16-row tile): one cross-lane move pairs them, lanes fg < 2 finish four activated columns
        // (v_permlane32_swap: lanes l and l + 32 trade two of their four values, so the lower half finishes columns 2, 3 and the upper half
        //  columns 0, 1 of the group - two VALU swaps per fragment and the SiLU work on all 64 lanes; four ds_bpermute + half the lanes idle cost
        //  the gate/up GEMM of the prefill 10 %)
        typedef typename ET<OT>::v2 O2;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int ol = wc * 32 + nb * 8 + (fg & 1) * 4 + (fg < 2 ? 2 : 0);     // first of this lane's two columns in the 128-wide activated tile
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const f32x4 v = {(float)acc[nb][mb][0], (float)acc[nb][mb][1], (float)acc[nb][mb][2], (float)acc[nb][mb][3]};
                // lower half gets (G2, U2) / (G3, U3), upper half (G0, U0) / (G1, U1)
                const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[0]), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[3]), __float_as_uint(v[1]), false, false);
                O2 o;
                o[0] = (OT)(rT<OT>(silu_f(rT<OT>(__uint_as_float(s0[0])))) * rT<OT>(__uint_as_float(s0[1])));
                o[1] = (OT)(rT<OT>(silu_f(rT<OT>(__uint_as_float(s1[0])))) * rT<OT>(__uint_as_float(s1[1])));
                *(O2*)(smem + (wr * 128 + mb * 16 + fr) * CLD + ol * 2) = o;
            }
        }
        __syncthreads();
        const int No = a.N >> 1, o0 = n0 >> 1;
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            const int c = it * 512 + tid, row = c >> 4, ch = c & 15;
            const int m = m0 + row, oc = o0 + ch * 8;
            if (m < a.M && oc < No) *(O8*)(C + (long)m * a.ldc + oc) = *(const O8*)(smem + row * CLD + ch * 16);
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ol = wc * 32 + q * 16 + fg * 4;           // column in the 128-wide activated tile
            int ng = n0 + wc * 64 + q * 32 + fg * 4; ng = ng + 19 < a.N ? ng : 0;   // gate columns ng .., up columns ng + 16 .. (clamped: unused beyond N)
            f32x4 sbg = {0.f, 0.f, 0.f, 0.f}, sbu = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KD::I8) { sbg = *(const f32x4*)(a.q.scb + ng); sbu = *(const f32x4*)(a.q.scb + ng + 16); }
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                O4 o;
                int m = m0 + wr * 128 + mb * 16 + fr; m = m < a.M ? m : a.M - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float g = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[2 * q][mb][j], m, ng + j, 0.f, rws[mb], sbg[j]), u = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[2 * q + 1][mb][j], m, ng + 16 + j, 0.f, rws[mb], sbu[j]);
                    o[j] = (OT)(rT<OT>(silu_f(g)) * u);
                }
                *(O4*)(smem + (wr * 128 + mb * 16 + fr) * CLD + ol * 2) = o;
            }
        }
        __syncthreads();
        const int No = a.N >> 1, o0 = n0 >> 1;
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            const int c = it * 512 + tid, row = c >> 4, ch = c & 15;
            const int m = m0 + row, oc = o0 + ch * 8;
            if (m < a.M && oc < No) *(O8*)(C + (long)m * a.ldc + oc) = *(const O8*)(smem + row * CLD + ch * 16);
        }
        return;
    }
    // fused partial RoPE (encoder q / k heads): a head is this wave's 64 columns; dims [0,16) are accumulator block 0 and their
    // rotation partners [16,32) block 1 of the same lane, so the rotate-half pair never leaves the registers
    // (compiled into the 16-bit QKV epilogue only: on top of the int8 dequantisation it makes the kernel spill inside its K loop)
    const bool rope = !KD::I8 && EPI == EPI_QKV_VT && a.rope_cs && (n0 + wc * 64) < a.rope_ncols;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int nl = wc * 64 + nb * 16 + fg * 4, n = n0 + nl;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias && n < a.N) {
            const f32x4 b4 = SBIAS ? *(const f32x4*)(sbias + nl) : *(const f32x4*)(a.bias + n);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
        }
        const int nc = n + 3 < a.N ? n : 0;                  // (clamped: columns beyond N are computed and dropped)
        f32x4 sb = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KD::I8) sb = *(const f32x4*)(a.q.scb + nc);
        if (rope && nb == 1) continue;                       // written together with block 0
        if constexpr (KD::I8) {
            // Outlier columns as an outer product.  When the 8 rows of this lane belong to one request (one outlier list; a 256-row tile
            // spans at most a few requests) the list is walked ONCE per 8 x 4 block: per column k, 1 index + 8 activations + 4 weights are
            // loaded for 32 products, instead of 2 loads per product in the per-element loop (prefill down_proj with ~360 outlier columns:
            // 1121 us per GEMM).  Every output element still adds its products in ascending k, each sum rounded to fp32: same bits.
            bool uni = rws[0].cnt > 0;
#pragma unroll
            for (int mb = 1; mb < 8; ++mb) uni = uni && rws[mb].g == rws[0].g;
            if (uni) {
                float v[8][4], a2[8][4];
                long xo[8];
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    int m = m0 + wr * 128 + mb * 16 + fr; m = m < a.M ? m : a.M - 1;
                    xo[mb] = (long)m * a.q.ldx16;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[mb][j] = rT<f16_t>(fmaf((float)acc[nb][mb][j], __fmul_rn(__fmul_rn(rws[mb].sa, sb[j]), MM_DEQUANT_CONST), bv[j]));
                        a2[mb][j] = 0.f;
                    }
                }
                const int cnt = rws[0].cnt;
                const int* lst = a.q.oc_list + (long)rws[0].g * a.q.oc_ld;
                const f16_t* xb = (const f16_t*)a.q.x16;
                // W[nc + j][k]: from the row-major matrix (rows K apart, k contiguous)
                // ... or (round 6: w_tiled without a k-major copy) from the fragment-tiled copy itself: the four columns nc .. nc + 3 lie in one 16-row group,
                // 16 bytes apart; the k part of the address is i8_tiled_k_off(k)
                const bool tg = a.w_tiled != 0;
                const int8_t* wb = tg ? (const int8_t*)a.W + i8_tiled_row_off(nc, a.K) : (const int8_t*)a.W + (long)nc * a.K;
                const int wsj = tg ? 16 : a.K;
                for (int i = 0; i < cnt; ++i) {
                    const int k = lst[i];
                    const int8_t* wp = wb + (tg ? i8_tiled_k_off(k) : (long)k);
                    float wd[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) wd[j] = rT<f16_t>(__fmul_rn(__fmul_rn((float)wp[j * wsj], sb[j]), INT8_DEQ_W));
#pragma unroll
                    for (int mb = 0; mb < 8; ++mb) {
                        const float xv = (float)xb[xo[mb] + k];
#pragma unroll
                        for (int j = 0; j < 4; ++j) a2[mb][j] = __fmaf_rn(xv, wd[j], a2[mb][j]);     // x, w are fp16 values: the product is exact in fp32
                    }
                }
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) {
                    O4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float l = rT<f16_t>(__fadd_rn(v[mb][j], a2[mb][j]));
                        o[j] = EPI == EPI_BIAS_GELU ? (OT)gelu_erf(l) : (OT)l;
                    }
                    *(O4*)(smem + (wr * 128 + mb * 16 + fr) * CLD + nl * 2) = o;
                }
                continue;
            }
        }
        float bv1[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 sb1 = {0.f, 0.f, 0.f, 0.f};
        if (rope && nb == 0) {
            if (a.bias) { const f32x4 b4 = SBIAS ? *(const f32x4*)(sbias + nl + 16) : *(const f32x4*)(a.bias + nc + 16); bv1[0] = b4[0]; bv1[1] = b4[1]; bv1[2] = b4[2]; bv1[3] = b4[3]; }
            if constexpr (KD::I8) sb1 = *(const f32x4*)(a.q.scb + nc + 16);
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            O4 o;
            int m = m0 + wr * 128 + mb * 16 + fr; m = m < a.M ? m : a.M - 1;
            if (rope && nb == 0) {
                const float* cs = a.rope_cs + (long)(m % a.rope_T) * 32 + fg * 4;
                const f32x4 c4 = *(const f32x4*)cs, s4 = *(const f32x4*)(cs + 16);
                O4 o2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x1 = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[0][mb][j], m, nc + j, bv[j], rws[mb], sb[j]);
                    const float x2 = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[1][mb][j], m, nc + 16 + j, bv1[j], rws[mb], sb1[j]);
                    o[j] = (OT)(rT<OT>(x1 * c4[j]) + rT<OT>(-x2 * s4[j]));
                    o2[j] = (OT)(rT<OT>(x2 * c4[j]) + rT<OT>(x1 * s4[j]));
                }
                *(O4*)(smem + (wr * 128 + mb * 16 + fr) * CLD + nl * 2) = o;
                *(O4*)(smem + (wr * 128 + mb * 16 + fr) * CLD + (nl + 16) * 2) = o2;
                continue;
            }
            if (EPI == EPI_BIAS_GELU) {
                if constexpr (LUT) {
                    if (a.gelu_lut) {
                        const unsigned short* lut = (const unsigned short*)(smem + LDS256_BYTES);
                        // Round 5: the table index in packed 16-bit arithmetic on the bf16 PAIRS v_cvt_pk_bf16_f32 produces (the round-4 form
                        // converted every value to bf16 and back, took the index from the float's bits and clamped / selected per element:
                        // ~14 VALU instructions per element, 8-10 us of a GELU tile's 16 us fixed cost).  A value outside the table
                        // (|x| < 2^-14 or >= 16: about 5e-5 of the elements) makes the wave redo the fragment the round-4 way: same bits.
                        typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
                        typedef __bf16 b16x2_t __attribute__((ext_vector_type(2)));
                        b16x2_t p01, p23;
                        p01[0] = (__bf16)((float)acc[nb][mb][0] + bv[0]); p01[1] = (__bf16)((float)acc[nb][mb][1] + bv[1]);
                        p23[0] = (__bf16)((float)acc[nb][mb][2] + bv[2]); p23[1] = (__bf16)((float)acc[nb][mb][3] + bv[3]);
                        const u16x2_t u01 = __builtin_bit_cast(u16x2_t, p01), u23 = __builtin_bit_cast(u16x2_t, p23);
                        const u16x2_t k7 = {0x7FFF, 0x7FFF}, kp = {GELU_LUT_E0 << 7, GELU_LUT_E0 << 7}, kh = {GELU_LUT_HALF, GELU_LUT_HALF};
                        const u16x2_t i01 = (u01 & k7) - kp, i23 = (u23 & k7) - kp;                   // index in the half table; wraps to >= HALF when outside
                        const u16x2_t mx = __builtin_elementwise_max(i01, i23);
                        const u16x2_t s01 = i01 + (u01 >> 15) * kh, s23 = i23 + (u23 >> 15) * kh;     // + HALF for negative values
                        const bool oob = mx[0] >= GELU_LUT_HALF || mx[1] >= GELU_LUT_HALF;
                        unsigned t0 = lut[s01[0]], t1 = lut[s01[1]], t2 = lut[s23[0]], t3 = lut[s23[1]];
                        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));                     // (the four reads stay unconditional and in flight together)
                        if (__builtin_amdgcn_ballot_w64(oob) == 0) {
                            typedef unsigned u32x2o_t __attribute__((ext_vector_type(2)));
                            u32x2o_t ov; ov[0] = t0 | (t1 << 16); ov[1] = t2 | (t3 << 16);
                            *(u32x2o_t*)(smem + (wr * 128 + mb * 16 + fr) * CLD + nl * 2) = ov;
                            continue;
                        }
                        float l[4]; unsigned t[4]; int idx[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            l[j] = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[nb][mb][j], m, nc + j, bv[j], rws[mb], sb[j]);                  // a bf16 value
                            idx[j] = gelu_lut_index(l[j]);
                            t[j] = lut[gelu_lut_slot(l[j], idx[j])];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(t[j]));
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_lut_value(l[j], idx[j], t[j]);
                        *(O4*)(smem + (wr * 128 + mb * 16 + fr) * CLD + nl * 2) = o;
                        continue;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_erf(gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[nb][mb][j], m, nc + j, bv[j], rws[mb], sb[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (OT)gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[nb][mb][j], m, nc + j, bv[j], rws[mb], sb[j]);   // RESID: the linear's own output; R is added below
            }
            *(O4*)(smem + (wr * 128 + mb * 16 + fr) * CLD + nl * 2) = o;
        }
    }
    __syncthreads();
    if constexpr (RPRE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 16; ++it) asm volatile("" : "+v"(rpre[it]));
    }
#pragma unroll RPRE ? 16 : 4
    for (int it = 0; it < 16; ++it) {
        const int c = it * 512 + tid, row = c >> 5, ch = c & 31;
        const int m = m0 + row, n = n0 + ch * 8;
        if (m < a.M && n < a.N) {
            O8 v = *(const O8*)(smem + row * CLD + ch * 16);
            if constexpr (KD::I8 && EPI == EPI_QKV_VT) {
                // int8 q / k tile: the encoder's partial RoPE on the way out (the register form of the 16-bit kinds is over the VGPR budget
                // beside the dequantisation): a head is 64 columns = 8 chunks; chunk 0 / 1 hold dims [0,16), their partners [16,32) sit two
                // chunks on; this thread writes both.  Same arithmetic as rope_enc_kernel.
                if (a.rope_cs && n0 < a.rope_ncols) {
                    const int hl = (ch & 7) * 8;
                    if (hl >= 16 && hl < 32) continue;
                    if (hl < 16) {
                        const O8 v2 = *(const O8*)(smem + row * CLD + (ch + 2) * 16);
                        const float* cs = a.rope_cs + (long)(m % a.rope_T) * 32 + hl;
                        const f32x4 c0 = *(const f32x4*)cs, c1 = *(const f32x4*)(cs + 4), s0 = *(const f32x4*)(cs + 16), s1 = *(const f32x4*)(cs + 20);
                        O8 o1, o2;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float x1 = (float)v[j], x2 = (float)v2[j], cc = j < 4 ? c0[j & 3] : c1[j & 3], ss = j < 4 ? s0[j & 3] : s1[j & 3];
                            o1[j] = (OT)(rT<OT>(x1 * cc) + rT<OT>(-x2 * ss));
                            o2[j] = (OT)(rT<OT>(x2 * cc) + rT<OT>(x1 * ss));
                        }
                        *(O8*)(C + (long)m * a.ldc + n) = o1;
                        *(O8*)(C + (long)m * a.ldc + n + 16) = o2;
                        continue;
                    }
                }
            }
            if constexpr (DEFER) {
                if (smem[LDS256_BYTES + row]) { *(O8*)((OT*)a.q.defer_out + (long)blockIdx.z * a.strideC + (long)m * a.ldc + n) = v; continue; }
            }
            if (EPI == EPI_BIAS_RESID) {
                O8 rv;
                if constexpr (RPRE) rv = __builtin_bit_cast(O8, rpre[it]); else rv = *(const O8*)(R + (long)m * a.ldr + n);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (OT)((float)v[j] + (float)rv[j]);
            }
            *(O8*)(C + (long)m * a.ldc + n) = v;
        }
    }
    G256_STAMP(3);
}


template <typename KD, int EPI, bool STG> static void launch256v(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = LDS256_BYTES + ((EPI == EPI_BIAS_GELU && std::is_same<KD, KBF16>::value) ? GELU_LUT_N * 2 : 0) + ((KD::I8 && EPI == EPI_BIAS_RESID) ? 256 : 0) + (KD::I8 ? 0 : 1024);
    ensure_dyn_lds((const void*)gemm256_kernel<KD, EPI, STG>, LDS);
    const int tilesM = (a.M + T256 - 1) / T256, tilesN = (a.N + T256 - 1) / T256;
    hipLaunchKernelGGL((gemm256_kernel<KD, EPI, STG>), dim3(tilesM * tilesN, 1, a.batch > 0 ? a.batch : 1), dim3(512), LDS, s, a);
}
template <int EPI> static void launch256(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    // raster group height: 8 M tiles per group; 2 when the matrix is at most 5 tiles wide (out_proj / fc2 of the encoder, N = 1280: +3-6 %
    // at M = 48000, tools/ab_gemm_raster.py); the option overrides both
    if (a.raster_gm <= 0) a.raster_gm = g_opts.gemm256_gm != 8 ? g_opts.gemm256_gm : ((a.N + T256 - 1) / T256 <= 5 ? 2 : 8);
    if (a.q.sca) launch256v<KI8, EPI, true>(a, s);
    else if (a.dt == DT_F16) launch256v<KF16, EPI, true>(a, s);
    else if (g_opts.gemm256_stagger) launch256v<KBF16, EPI, true>(a, s);
    else launch256v<KBF16, EPI, false>(a, s);                  // the un-staggered schedule is kept for bf16 experiments only
}

bool gemm256_eligible(const GemmArgs& a, int epi) {
    const int tbk = a.q.sca ? TBKB : TBKB / 2;                 // K elements per tile
    if (a.K % tbk || a.K / tbk < 4) return false;
    if (a.M < 512 || a.N < 256) return false;
    if (a.N % 16 || a.ldc % 8 || (epi == EPI_BIAS_RESID && a.ldr % 8)) return false;   // 16-byte row pieces in the staged epilogue
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
