// gemm256p_kernel: the persistent form of gemm256_kernel (gemm256.hip) for the 16-bit kinds.  Same 256x256x64 tile, same LDS-DMA ring,
// same staggered K loop and the same arithmetic per output element; what changes is what happens BETWEEN tiles.
//
// Why (round 3, tools/bench_gemm_k.py at M = 48000): the K loop itself runs at 1.33-1.35 PF/s = the plain-HIP 256^2 8-phase template of the
// CDNA guide, but every tile pays a fixed 7.5 us (bias) / 16 us (GELU) / 12-18 us (residual) on top of it: block launch, a prologue
// whose two K tiles of DMA have nothing to hide behind, the staged epilogue, and the output of all 256 CUs hitting HBM in one burst
// (33 MB per round) because the rounds run in lockstep.  At K = 1280 a tile's K loop is 32 us, so that is 19-36 % of the GEMM.
//   * one block per CU loops over its tiles (virtual block id = blockIdx.x + i * gridDim.x through the same XCD-aware grouped raster);
//   * the next tile's first two K tiles are requested into the ring BEFORE the finished tile's epilogue runs;
//   * the epilogue therefore cannot stage the tile in the ring: it goes through the 32 KiB behind it in four 64-row passes (eight
//     32-row passes where the GELU table lives there too); the stores of a pass are not waited for - they drain under the next tile's K
//     loop (stores count in vmcnt, but everything the K loop waits for with a counted vmcnt is OLDER than they are or is issued
//     after them, so the counted waits stay correct: conservative at the first K tile, exact afterwards);
//   * the GELU table is copied to LDS once per block instead of once per tile.
// The arithmetic of every epilogue is copied from gemm256.hip: a result must not depend on which kernel computed it.
//
// MEASURED (round 3, tools/ab_gemm_persist.py, one MI355X, M = 48000): SLOWER than one block per tile in every shape - out_proj 200 vs 170 us,
// fc1 + GELU 739 vs 686, fc2 564 vs 552, QKV + V^T + RoPE 957 vs 453 (that instantiation spills 1.1 KB per lane) - so it is OFF by default
// (option "gemm256_persist").  What the overlap buys (no block launch, the next prologue's DMA latency, stores draining under the next
// K loop: about 3 us per tile) is less than what the passes cost: four staging passes with two barriers each, and in every pass only the
// four waves that own the pass's rows convert / look up / write while the other four wait.  The fixed cost of a tile is mostly the
// epilogue's own VALU + LDS work, which the same waves have to do either way; hiding it under MFMA work needs a second accumulator set
// (256 more VGPRs) or 2 blocks per CU (the 128 KiB ring does not allow it).  Parity tests pass with it on.
#include <type_traits>

#include "common.h"
#include "int8_util.h"

#define T256 256
#define TBKB 128
#define HT_BYTES (128 * TBKB)
#define TILE_BYTES (4 * HT_BYTES)
#define RING_BYTES (2 * TILE_BYTES)          // 128 KiB
#define SLOT_A0 0
#define SLOT_A1 1
#define SLOT_B0 2
#define SLOT_B1 3
#define PBARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
template <int N> __device__ __forceinline__ void pwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename KD, int EPI>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmArgs a) {
    typedef typename KD::elem ET_; typedef typename KD::out OT; typedef typename KD::frag Frag; typedef typename KD::acc Acc;
    typedef typename ET<OT>::v4 O4; typedef typename ET<OT>::v8 O8;
    static_assert(!KD::I8, "16-bit kinds only");
    constexpr int CE = 8, TBK = 64;
    constexpr bool LUT = EPI == EPI_BIAS_GELU && std::is_same<KD, KBF16>::value;
    constexpr int LUT_BYTES = LUT ? GELU_LUT_N * 2 : 0;
    constexpr int SR = LUT ? 32 : 64;            // rows per staging pass (SwiGLU / QKV + V^T epilogues: block-wide passes)
    constexpr int NPASS = 256 / SR;
    // bias / GELU / residual epilogues (round 5): every wave transposes ITS 128 x 64 outputs through a private 16-row scratch (WS_PITCH bytes per
    // row) - no block barrier in the epilogue, all eight waves work at once
    constexpr bool PERWAVE = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID;
    constexpr int WS_PITCH = 144, WS_BYTES = 16 * WS_PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ring | GELU table | staging (SR rows x 512 B, XOR-swizzled 16-B chunks; or 8 wave scratches)
    char* const stg = smem + RING_BYTES + LUT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int tilesM = (a.M + T256 - 1) / T256, tilesN = (a.N + T256 - 1) / T256;
    const int nt = tilesM * tilesN;
    const int GMr = a.raster_gm > 0 ? a.raster_gm : 8;
    auto tile_of = [&](int vb, int& tm, int& tn) {
        const int q = nt >> 3, r = nt & 7, xcd = vb & 7, loc = vb >> 3;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int gsz = GMr * tilesN, g = id / gsz, first = g * GMr;
        const int gm = min(GMr, tilesM - first), in = id - g * gsz;
        tm = first + in % gm;
        tn = in / gm;
    };
    const ET_* A = (const ET_*)a.A;
    OT* C = (OT*)a.C;
    const OT* R = (EPI == EPI_BIAS_RESID) ? (const OT*)a.R : nullptr;

    const int lr8 = lane >> 3, lc = (lane & 7) ^ lr8;
    const ET_* srcA[2][2];
    const ET_* srcB[2][2];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lrow = (wid * 2 + i) * 8 + lr8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int m = m0 + (lrow >> 6) * 128 + h * 64 + (lrow & 63); m = m < a.M ? m : a.M - 1;
                int n = n0 + (lrow >> 5) * 64 + h * 32 + (lrow & 31); n = n < a.N ? n : a.N - 1;
                srcA[h][i] = A + (long)m * a.lda + lc * CE;
                srcB[h][i] = (const ET_*)a.W + (long)n * a.K + lc * CE;
            }
        }
    };
    auto dma = [&](const ET_* const (&src)[2], int k0, int buf, int slot) {
        char* dst = smem + buf * TILE_BYTES + slot * HT_BYTES + wid * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    auto prologue = [&]() {       // K tiles 0 and 1, issue order = consumption order (A0,B0 | B1 | A1)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            dma(srcA[0], t * TBK, t, SLOT_A0); dma(srcB[0], t * TBK, t, SLOT_B0);
            dma(srcB[1], t * TBK, t, SLOT_B1);
            dma(srcA[1], t * TBK, t, SLOT_A1);
        }
    };

    if constexpr (LUT) {          // once per block; older than every operand DMA, so the first counted wait covers it
        if (wid == 0 && a.gelu_lut) {
#pragma unroll
            for (int i = 0; i < GELU_LUT_N * 2 / 1024; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)a.gelu_lut + i * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + RING_BYTES + i * 1024), 16, 0, 0);
        }
    }

    Acc acc[4][8];   // [n-block][m-block]
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
    auto read_b = [&](int buf, int half, Frag (&b)[2][2]) {
        const char* s = smem + buf * TILE_BYTES + (half ? SLOT_B1 : SLOT_B0) * HT_BYTES;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int row = wc * 32 + ni * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) b[ni][kk] = *(const Frag*)(s + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
        }
    };
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
    using I0 = std::integral_constant<int, 0>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I10 = std::integral_constant<int, 10>;
    using I12 = std::integral_constant<int, 12>;
    using IM1 = std::integral_constant<int, -1>;
    using Yes = std::true_type;
    using No = std::false_type;
    const int nk = a.K / TBK;   // >= 4
    auto lgkm0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
    // the staggered K tile of gemm256.hip (waves w and w + 4 half a phase apart, 7 barriers per K tile)
    auto stile = [&](auto grp, auto vt, int kt, auto w1, auto w2, auto w0n, auto issue, bool first) {
        constexpr bool G0 = decltype(grp)::value;
        constexpr int W1 = decltype(w1)::value, W2 = decltype(w2)::value, W0N = decltype(w0n)::value;
        constexpr bool ISSUE = decltype(issue)::value;
        const int buf = kt & 1, kn = (kt + 2) * TBK;
        PBARRIER();                                                                  // slot 0
        if (G0) { read_a(buf, 0); read_b(buf, 0, b0); lgkm0(); } else if (!first) quad(vt, 1, 0, b0);
        PBARRIER();                                                                  // slot 1
        if (G0) quad(vt, 0, 0, b0); else { read_a(buf, 0); read_b(buf, 0, b0); lgkm0(); }
        pwait_vm<W1>();
        PBARRIER();                                                                  // slot 2
        if (ISSUE) { dma(srcA[0], kn, buf, SLOT_A0); dma(srcB[0], kn, buf, SLOT_B0); }
        if (G0) { read_b(buf, 1, b1); lgkm0(); } else quad(vt, 0, 0, b0);
        PBARRIER();                                                                  // slot 3
        if (G0) quad(vt, 0, 1, b1); else { read_b(buf, 1, b1); lgkm0(); }
        pwait_vm<W2>();
        PBARRIER();                                                                  // slot 4
        if (ISSUE) dma(srcB[1], kn, buf, SLOT_B1);
        if (G0) { read_a(buf, 1); lgkm0(); } else quad(vt, 0, 1, b1);
        PBARRIER();                                                                  // slot 5
        if (G0) quad(vt, 1, 1, b1); else { read_a(buf, 1); lgkm0(); }
        PBARRIER();                                                                  // slot 6
        if (ISSUE) dma(srcA[1], kn, buf, SLOT_A1);
        if (G0) quad(vt, 1, 0, b0); else quad(vt, 1, 1, b1);
        if (W0N >= 0) pwait_vm<(W0N >= 0 ? W0N : 0)>();
    };
    auto run = [&](auto grp, auto vt) {
        pwait_vm<12>();                                                               // D0(0) landed
        int kt = 0;
        for (; kt < nk - 2; ++kt) stile(grp, vt, kt, I10{}, I12{}, I12{}, Yes{}, kt == 0);
        stile(grp, vt, kt, I10{}, I8{}, I4{}, No{}, false);
        ++kt;
        stile(grp, vt, kt, I2{}, I0{}, IM1{}, No{}, false);
        if (!decltype(grp)::value) quad(vt, 1, 0, b0);                               // waves 4-7: last K tile's quadrant 3
    };

    // staging image: SR rows x 512 B (256 16-bit columns; SWIGLU: 128 columns = 256 B used), 16-byte chunk c of row r at chunk c ^ (r & 31)
    auto stg_addr = [&](int r, int col_elems) -> char* {
        const int byte = col_elems * 2, chunk = byte >> 4;
        return stg + r * 512 + (((chunk ^ (r & 31)) << 4) | (byte & 15));
    };

    int tm, tn;
    int vb = blockIdx.x;
    if (vb >= nt) return;
    tile_of(vb, tm, tn);
    set_src(tm * T256, tn * T256);
    prologue();
    for (;;) {
        const int m0 = tm * T256, n0 = tn * T256;
        const bool vtile = (EPI == EPI_QKV_VT) && (n0 >= a.n_split);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;
        if constexpr (EPI == EPI_QKV_VT) {
            if (vtile) { if (wid < 4) run(Yes{}, Yes{}); else run(No{}, Yes{}); }
            else { if (wid < 4) run(Yes{}, No{}); else run(No{}, No{}); }
        } else { if (wid < 4) run(Yes{}, No{}); else run(No{}, No{}); }
        __syncthreads();                         // every fragment of the last K tile has been read: the ring may be refilled

        // ---- this tile's bias (the 4 x 4 columns of this lane), requested by asm loads so that their wait can be COUNTED: the compiler's own
        // bookkeeping would wait vmcnt(0) at the first use of an ordinary load, i.e. for the next tile's operand DMAs issued just below
        f32x4 b4[4];
        const bool has_bias = a.bias != nullptr && !(EPI == EPI_QKV_VT && vtile) && EPI != EPI_SWIGLU;
        if (has_bias) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int n = n0 + wc * 64 + nb * 16 + fg * 4; n = n + 3 < a.N ? n : 0;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b4[nb]) : "v"(a.bias + n) : "memory");
            }
        }
        // ---- residual rows of this wave's 128 x 64 outputs (per-wave epilogue): 16 x 16 bytes per lane, asm loads for the same reason
        i32x4 rres[PERWAVE && EPI == EPI_BIAS_RESID ? 16 : 1];
        if constexpr (PERWAVE && EPI == EPI_BIAS_RESID) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int m = m0 + wr * 128 + (q >> 1) * 16 + (lane >> 3) + 8 * (q & 1); m = m < a.M ? m : a.M - 1;
                int n = n0 + wc * 64 + (lane & 7) * 8; n = n < a.N ? n : 0;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rres[q]) : "v"(R + (long)m * a.ldr + n) : "memory");
            }
        }
        // ---- the next tile's first two K tiles go out now; the epilogue below works beside them
        const int vbn = vb + gridDim.x;
        const bool more = vbn < nt;
        int tmn = 0, tnn = 0;
        if (more) { tile_of(vbn, tmn, tnn); set_src(tmn * T256, tnn * T256); prologue(); }
        if (has_bias) {
            if constexpr (PERWAVE && EPI == EPI_BIAS_RESID) { if (more) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
            else { if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+v"(b4[nb]));
        } else {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) b4[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

        // ---- epilogue.  acc[nb][mb][j] = D[n = n0 + wc*64 + nb*16 + fg*4 + j][m = m0 + wr*128 + mb*16 + fr]  (vtile: roles swapped)
        if (EPI == EPI_QKV_VT && vtile) {
            // staged transposed in passes of SR n-rows: row = n, columns = m; V^T[seg][n - n_split][t .. t+3] leaves in 8-byte pieces
            // pass p holds n-rows [SR * p, SR * p + SR): wave column wc = (SR * p) / 64, its n-blocks nb with (wc*64 + nb*16) in the pass
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (wc == (p * SR) / 64) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        if ((nb * 16) / SR != ((p * SR) % 64) / SR) continue;
                        const int nl = wc * 64 + nb * 16 + fr, n = n0 + nl;
                        const float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
                        for (int mb = 0; mb < 8; ++mb) {
                            O4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (OT)rT<OT>((float)acc[nb][mb][j] + bv);
                            *(O4*)stg_addr(nl - p * SR, wr * 128 + mb * 16 + fg * 4) = o;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < SR * 64 / 512; ++it) {
                    const int c = it * 512 + tid, nl = c >> 6, mc = c & 63;
                    const int m = m0 + mc * 4, n = n0 + p * SR + nl;
                    if (m < a.M && n < a.N) {
                        const int seg = m / a.seg_T, t = m - seg * a.seg_T;
                        *(O4*)((OT*)a.Vt + (long)seg * a.vt_seg_stride + (long)(n - a.n_split) * a.vt_ld + t) = *(const O4*)stg_addr(nl, mc * 4);
                    }
                }
                __syncthreads();
            }
        } else if (EPI == EPI_SWIGLU) {
            const int No = a.N >> 1, o0 = n0 >> 1;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (wr == (p * SR) / 128) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int ol = wc * 32 + q * 16 + fg * 4;           // column in the 128-wide activated tile
#pragma unroll
                        for (int h = 0; h < SR / 16; ++h) {
                            const int mb = ((p * SR) % 128) / 16 + h;
                            O4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float g = rT<OT>((float)acc[2 * q][mb][j]), u = rT<OT>((float)acc[2 * q + 1][mb][j]);
                                o[j] = (OT)(rT<OT>(silu_f(g)) * u);
                            }
                            *(O4*)stg_addr(h * 16 + fr, ol) = o;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < SR * 16 / 512; ++it) {
                    const int c = it * 512 + tid, row = c >> 4, ch = c & 15;
                    const int m = m0 + p * SR + row, oc = o0 + ch * 8;
                    if (m < a.M && oc < No) *(O8*)(C + (long)m * a.ldc + oc) = *(const O8*)stg_addr(row, ch * 8);
                }
                __syncthreads();
            }
        } else if constexpr (PERWAVE) {
            // acc[nb][mb][j] = D[n = n0 + wc*64 + nb*16 + fg*4 + j][m = m0 + wr*128 + mb*16 + fr].  Pass mb: the wave's 16 rows x 64 columns go
            // through its scratch (lane (fr, fg) writes 8 bytes at row fr, column nb*16 + fg*4) and leave as 128-byte row pieces (8 lanes x 16 B).
            // LDS operations of one wave execute in order, so neither the read-after-write inside a pass nor the write-after-read between passes
            // needs a barrier.  Residual rows were requested before the next tile's operand DMAs and are waited for with a counted vmcnt
            // (they are older than the DMAs; the stores of earlier passes are younger): the K loop's DMAs stay in flight under the epilogue.
            char* const ws = stg + wid * WS_BYTES;
            const int rrow = lane >> 3, rch = lane & 7;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const float bv[4] = {b4[nb][0], b4[nb][1], b4[nb][2], b4[nb][3]};
                    O4 o;
                    if (EPI == EPI_BIAS_GELU) {
                        bool done = false;
                        if constexpr (LUT) {
                            if (a.gelu_lut) {
                                const unsigned short* lut = (const unsigned short*)(smem + RING_BYTES);
                                float l[4]; unsigned t[4]; int idx[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    l[j] = rT<OT>((float)acc[nb][mb][j] + bv[j]);
                                    idx[j] = gelu_lut_index(l[j]);
                                    t[j] = lut[gelu_lut_slot(l[j], idx[j])];
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(t[j]));
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_lut_value(l[j], idx[j], t[j]);
                                done = true;
                            }
                        }
                        if (!done) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_erf(rT<OT>((float)acc[nb][mb][j] + bv[j]));
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (OT)rT<OT>((float)acc[nb][mb][j] + bv[j]);   // RESID: the linear's own output; R is added below
                    }
                    *(O4*)(ws + fr * WS_PITCH + (nb * 16 + fg * 4) * 2) = o;
                }
                if (EPI == EPI_BIAS_RESID) {
                    // residual pieces of this pass have landed: all but the (14 - 2 mb) younger residual loads, the next tile's 16 DMAs and the 2 mb stores so far
                    if (more) asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = rrow + 8 * i;
                    const int m = m0 + wr * 128 + mb * 16 + row, n = n0 + wc * 64 + rch * 8;
                    O8 v = *(const O8*)(ws + row * WS_PITCH + rch * 16);
                    if (EPI == EPI_BIAS_RESID) {
                        asm volatile("" : "+v"(rres[mb * 2 + i]));
                        const O8 rv = __builtin_bit_cast(O8, rres[mb * 2 + i]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (OT)((float)v[j] + (float)rv[j]);
                    }
                    if (m < a.M && n < a.N) *(O8*)(C + (long)m * a.ldc + n) = v;
                }
            }
        } else {
            // fused partial RoPE (encoder q / k heads): dims [0,16) are accumulator block 0 and their rotation partners [16,32) block 1 of the
            // same lane (gemm256.hip)
            const bool rope = EPI == EPI_QKV_VT && a.rope_cs && (n0 + wc * 64) < a.rope_ncols;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                if (wr == (p * SR) / 128) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const int nl = wc * 64 + nb * 16 + fg * 4;
                        const float bv[4] = {b4[nb][0], b4[nb][1], b4[nb][2], b4[nb][3]};
                        if (rope && nb == 1) continue;                       // written together with block 0
                        const float bv1[4] = {b4[1][0], b4[1][1], b4[1][2], b4[1][3]};   // (rope, nb == 0: the bias of the rotation partners = block 1's)
#pragma unroll
                        for (int h = 0; h < SR / 16; ++h) {
                            const int mb = ((p * SR) % 128) / 16 + h;
                            const int rl = h * 16 + fr;
                            O4 o;
                            int m = m0 + wr * 128 + mb * 16 + fr; m = m < a.M ? m : a.M - 1;
                            if (rope && nb == 0) {
                                const float* cs = a.rope_cs + (long)(m % a.rope_T) * 32 + fg * 4;
                                const f32x4 c4 = *(const f32x4*)cs, s4 = *(const f32x4*)(cs + 16);
                                O4 o2;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float x1 = rT<OT>((float)acc[0][mb][j] + bv[j]);
                                    const float x2 = rT<OT>((float)acc[1][mb][j] + bv1[j]);
                                    o[j] = (OT)(rT<OT>(x1 * c4[j]) + rT<OT>(-x2 * s4[j]));
                                    o2[j] = (OT)(rT<OT>(x2 * c4[j]) + rT<OT>(x1 * s4[j]));
                                }
                                *(O4*)stg_addr(rl, nl) = o;
                                *(O4*)stg_addr(rl, nl + 16) = o2;
                                continue;
                            }
                            if (EPI == EPI_BIAS_GELU) {
                                if constexpr (LUT) {
                                    if (a.gelu_lut) {
                                        const unsigned short* lut = (const unsigned short*)(smem + RING_BYTES);
                                        float l[4]; unsigned t[4]; int idx[4];
#pragma unroll
                                        for (int j = 0; j < 4; ++j) {
                                            l[j] = rT<OT>((float)acc[nb][mb][j] + bv[j]);
                                            idx[j] = gelu_lut_index(l[j]);
                                            t[j] = lut[gelu_lut_slot(l[j], idx[j])];
                                        }
#pragma unroll
                                        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(t[j]));
#pragma unroll
                                        for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_lut_value(l[j], idx[j], t[j]);
                                        *(O4*)stg_addr(rl, nl) = o;
                                        continue;
                                    }
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_erf(rT<OT>((float)acc[nb][mb][j] + bv[j]));
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[j] = (OT)rT<OT>((float)acc[nb][mb][j] + bv[j]);   // RESID: the linear's own output; R is added below
                            }
                            *(O4*)stg_addr(rl, nl) = o;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < SR * 32 / 512; ++it) {
                    const int c = it * 512 + tid, row = c >> 5, ch = c & 31;
                    const int m = m0 + p * SR + row, n = n0 + ch * 8;
                    if (m < a.M && n < a.N) {
                        O8 v = *(const O8*)stg_addr(row, ch * 8);
                        if (EPI == EPI_BIAS_RESID) {
                            const O8 rv = *(const O8*)(R + (long)m * a.ldr + n);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = (OT)((float)v[j] + (float)rv[j]);
                        }
                        *(O8*)(C + (long)m * a.ldc + n) = v;
                    }
                }
                __syncthreads();
            }
        }
        if (!more) break;
        vb = vbn; tm = tmn; tn = tnn;
    }
}

template <typename KD, int EPI> static void launch256p_v(const GemmArgs& a, int cus, hipStream_t s) {
    constexpr bool LUT = EPI == EPI_BIAS_GELU && std::is_same<KD, KBF16>::value;
    constexpr bool PERWAVE = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID;
    constexpr int LDS = RING_BYTES + (PERWAVE ? (LUT ? GELU_LUT_N * 2 : 0) + 8 * 16 * 144 : (LUT ? GELU_LUT_N * 2 + 32 * 512 : 64 * 512));
    ensure_dyn_lds((const void*)gemm256p_kernel<KD, EPI>, LDS);
    const int tilesM = (a.M + T256 - 1) / T256, tilesN = (a.N + T256 - 1) / T256, nt = tilesM * tilesN;
    int grid = nt < cus ? nt : cus;
    if (grid >= 8) grid &= ~7;                   // the XCD-aware raster wants a multiple of 8 resident blocks
    hipLaunchKernelGGL((gemm256p_kernel<KD, EPI>), dim3(grid), dim3(512), LDS, s, a);
}
// persistent 256x256 GEMM for the 16-bit kinds; a: as launch_gemm256 (batch <= 1)
void launch_gemm256p(const GemmArgs& a0, int epi, int cus, hipStream_t s) {
    GemmArgs a = a0;
    if (a.raster_gm <= 0) a.raster_gm = g_opts.gemm256_gm != 8 ? g_opts.gemm256_gm : ((a.N + T256 - 1) / T256 <= 5 ? 2 : 8);
#define P256(KD) do { switch (epi) { \
        case EPI_BIAS: launch256p_v<KD, EPI_BIAS>(a, cus, s); break; \
        case EPI_BIAS_GELU: launch256p_v<KD, EPI_BIAS_GELU>(a, cus, s); break; \
        case EPI_BIAS_RESID: launch256p_v<KD, EPI_BIAS_RESID>(a, cus, s); break; \
        case EPI_SWIGLU: launch256p_v<KD, EPI_SWIGLU>(a, cus, s); break; \
        case EPI_QKV_VT: launch256p_v<KD, EPI_QKV_VT>(a, cus, s); break; } } while (0)
    if (a.dt == DT_F16) P256(KF16); else P256(KBF16);
#undef P256
}
