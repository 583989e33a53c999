// bf16 MFMA GEMMs for gfx950.
//
//   gemm_kernel    C[M][N] = A[M][K] . W[N][K]^T (+ epilogue)      -- encoder / projector / prefill linears,
//                  conv stem as im2col-free GEMM (torch nn.Linear layout, so both
//                  operands are K-contiguous and every MFMA fragment is one 16-byte LDS read)
//   skinny_kernel  partial[ks][M<=64][N] = X . W^T over a K slice  -- decode-step weight streaming (K10/K11)
//
// gemm_kernel: 128x128x64 block tile, 4 waves (2x2), 64x64 per wave as 4x4 v_mfma_f32_16x16x32_bf16,
// operands staged HBM->LDS with global_load_lds_dwordx4 (16 B/lane, lane-linear LDS image), XOR
// swizzle applied on the *source* address and again on the fragment read (chunk ^= row & 7) so the
// ds_read_b128 fragment reads are bank-conflict free; two LDS stages, next tile's DMA in flight
// under the current tile's MFMAs; one barrier per K step.  MFMA roles are swapped (A-operand = W
// rows, B-operand = activation rows) so each lane ends up with 4 consecutive output columns of one
// row and the epilogue stores 8-byte bf16 quads.
#include <type_traits>

#include "common.h"
#include "int8_util.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES ((BM + BN) * BK * 2)

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// NS = stages of the operand ring.  2: one k-step in flight, two blocks per CU hide each other's load latency (large grids).  4: three k-steps in
// flight with counted waits, one block per CU - for grids that do not fill the chip anyway (the <= 128 ragged rows a prefill GEMM cuts off its
// 256x256 launch: 16 blocks whose k-step was one full L2 round trip, 0.54 us, with NS = 2).  Same k order per accumulator: same bits.
// WM x WN = 16x16 MFMA tiles per wave (4 waves as 2 x 2): block tile (32 WM) x (32 WN).  4 x 4 = 128 x 128 is the general kernel; 1 x 2 =
// 32 x 64 is for problems of a few hundred rows (the ragged tail rows, the streaming partials): such a GEMM is bound by how many CUs pull
// operands (~65 GB/s each), and 16 tiles of 128 x 128 leave 240 CUs idle - 52 us for the 128 x 2048 x 6144 tail of a prefill down_proj.
// Every output element still sums its k-blocks of 32 in ascending order on the same MFMA instruction: the tile shape changes no bit.
template <typename KD, int EPI, int NS = 2, int WM = 4, int WN = 4>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void gemm_kernel(GemmArgs a) {
    constexpr int TBM = 32 * WM, TBN = 32 * WN, TSTAGE = (TBM + TBN) * BK * 2;
    static_assert(EPI != EPI_SWIGLU || WN % 2 == 0, "gate / up pairs of 16-column groups per wave");
    typedef typename KD::elem ET_; typedef typename KD::out OT; typedef typename KD::frag Frag; typedef typename KD::acc Acc;
    typedef typename ET<OT>::v4 O4;
    constexpr int EB = sizeof(ET_), CE = 16 / EB, BKE = 128 / EB;   // bytes per element, elements per 16-B chunk / per 128-B tile row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    // ---- tile id: XCD-aware (blocks b, b+8, ... share an L2) + grouped raster (8 tile-rows per group)
    const int tilesM = (a.M + TBM - 1) / TBM, tilesN = (a.N + TBN - 1) / TBN;
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
    const int m0 = tm * TBM, n0 = tn * TBN;
    const ET_* A = (const ET_*)a.A + (long)blockIdx.z * a.strideA;
    OT* C = (OT*)a.C + (long)blockIdx.z * a.strideC;
    const OT* R = (EPI == EPI_BIAS_RESID) ? (const OT*)a.R + (long)blockIdx.z * a.strideR : nullptr;

    // ---- per-lane DMA source pointers (4 row groups of 8 rows per wave, per operand)
    const int lr = lane >> 3, lp = lane & 7, lc = lp ^ lr;  // LDS row-in-group, physical chunk, logical chunk
    const ET_* srcA[WM];                                     // (WM / WN row groups of 8 rows per wave and operand)
    const ET_* srcW[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        int ra = m0 + (wid * WM + i) * 8 + lr; ra = ra < a.M ? ra : a.M - 1;
        srcA[i] = A + (long)ra * a.lda + lc * CE;
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        int rw = n0 + (wid * WN + i) * 8 + lr; rw = rw < a.N ? rw : a.N - 1;
        srcW[i] = (const ET_*)a.W + (long)rw * a.K + lc * CE;
        if (a.w_tiled) {
            // fragment-tiled W (GemmArgs.w_tiled): piece wid * WN + i of the stage = (row tile p >> 1, k-step p & 1), 1 KiB contiguous, lane-linear
            const int p = wid * WN + i;
            int nb = n0 + (p >> 1) * 16; nb = nb + 16 <= a.N ? nb : a.N - 16;
            srcW[i] = (const ET_*)a.W + ((long)(nb >> 4) * (a.K / (BKE / 2)) + (p & 1)) * (BKE / 2 * 16) + lane * CE;
        }
    }
    const int kmulW = a.w_tiled ? 16 : 1;
    auto stage_load = [&](int stage, int k0) {
        char* sA = smem + stage * TSTAGE;
        char* sB = sA + TBM * BK * 2;
#pragma unroll
        for (int i = 0; i < WM; ++i) glds16(srcA[i] + k0, sA + (wid * WM + i) * 1024);
#pragma unroll
        for (int i = 0; i < WN; ++i) glds16(srcW[i] + k0 * kmulW, sB + (wid * WN + i) * 1024);
    };

    Acc acc[WN][WM];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;

    const bool vtile = (EPI == EPI_QKV_VT) && (n0 >= a.n_split);
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = a.K / BKE;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) stage_load(p, min(p, nk - 1) * BKE);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt % NS;
        // stage kt has landed when at most NS - 2 later stage loads (WM + WN DMA instructions each) are still in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (WM + WN)) : "memory");
        __syncthreads();
        // the slot consumed in step kt - 1 is free for step kt + NS - 1 (past the end: the last block again, into a slot nobody reads - keeps
        // the wait counts constant)
        if (NS > 2 || kt + 1 < nk) stage_load((kt + NS - 1) % NS, min(kt + NS - 1, nk - 1) * BKE);
        const char* sA = smem + cur * TSTAGE;
        const char* sB = sA + TBM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            Frag xf[WM], wf[WN];
            const int c = kk * 4 + fg;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int rx = wr * (WM * 16) + i * 16 + fr;
                xf[i] = *(const Frag*)(sA + rx * 128 + ((c ^ (rx & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                const int rw = wc * (WN * 16) + i * 16 + fr;
                wf[i] = *(const Frag*)(sB + (a.w_tiled ? ((wc * WN + i) * 2 + kk) * 1024 + lane * 16 : rw * 128 + ((c ^ (rw & 7)) << 4)));
            }
            if (vtile) {
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) acc[ni][mi] = KD::mfma(xf[mi], wf[ni], acc[ni][mi]);
            } else {
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) acc[ni][mi] = KD::mfma(wf[ni], xf[mi], acc[ni][mi]);
            }
        }
    }

    // ---- epilogue
    if (vtile) {
        // acc[ni][mi][j] = D[m = mrow + j][n = ncol]; V^T[seg][n - n_split][t .. t+3]
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int n = n0 + wc * (WN * 16) + ni * 16 + fr;
            const float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                const int m = m0 + wr * (WM * 16) + mi * 16 + fg * 4;
                if (m < a.M && n < a.N) {
                    const int seg = m / a.seg_T, t = m - seg * a.seg_T;
                    O4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (OT)gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[ni][mi][j], m + j, n, bv, I8Row{0.f, 0, 0, false}, 0.f);   // (16-bit kinds only)
                    *(O4*)((OT*)a.Vt + (long)seg * a.vt_seg_stride + (long)(n - a.n_split) * a.vt_ld + t) = o;
                }
            }
        }
        return;
    }
    if (EPI == EPI_SWIGLU && a.gu8) {
        // gate / up in 8-row groups (GemmArgs.gu8): lanes fg < 2 hold four gate columns of a 16-row tile, lanes fg >= 2 their up partners
        // (v_permlane32_swap: the lower half finishes columns 2, 3 of its group of four, the upper half columns 0, 1 - see gemm256.hip)
        typedef typename ET<OT>::v2 O2;
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            const int oc = ((n0 + wc * (WN * 16) + ni * 16) >> 1) + (fg & 1) * 4 + (fg < 2 ? 2 : 0);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                const int m = m0 + wr * (WM * 16) + mi * 16 + fr;
                const f32x4 v = {(float)acc[ni][mi][0], (float)acc[ni][mi][1], (float)acc[ni][mi][2], (float)acc[ni][mi][3]};
                const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[0]), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[3]), __float_as_uint(v[1]), false, false);
                if (m < a.M && (n0 + wc * (WN * 16) + ni * 16) < a.N) {
                    O2 o;
                    o[0] = (OT)(rT<OT>(silu_f(rT<OT>(__uint_as_float(s0[0])))) * rT<OT>(__uint_as_float(s0[1])));
                    o[1] = (OT)(rT<OT>(silu_f(rT<OT>(__uint_as_float(s1[0])))) * rT<OT>(__uint_as_float(s1[1])));
                    *(O2*)(C + (long)m * a.ldc + oc) = o;
                }
            }
        }
        return;
    }
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int q = 0; q < WN / 2; ++q) {
            const int oc = ((n0 + wc * (WN * 16)) >> 1) + q * 16 + fg * 4;
            const int ng = n0 + wc * (WN * 16) + q * 32 + fg * 4;    // gate columns ng .. ng+3, up columns ng+16 ..
            f32x4 sbg = {0.f, 0.f, 0.f, 0.f}, sbu = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KD::I8) if (ng + 19 < a.N) { sbg = *(const f32x4*)(a.q.scb + ng); sbu = *(const f32x4*)(a.q.scb + ng + 16); }
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                const int m = m0 + wr * (WM * 16) + mi * 16 + fr;
                if (m < a.M && (n0 + wc * (WN * 16) + q * 32) < a.N) {
                    const I8Row rw = i8_row<KD>(a, m);
                    O4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[2 * q][mi][j], m, ng + j, 0.f, rw, sbg[j]), u = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[2 * q + 1][mi][j], m, ng + 16 + j, 0.f, rw, sbu[j]);
                        o[j] = (OT)(rT<OT>(silu_f(g)) * u);
                    }
                    *(O4*)(C + (long)m * a.ldc + oc) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int n = n0 + wc * (WN * 16) + ni * 16 + fg * 4;
        if (n >= a.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const f32x4 b4 = *(const f32x4*)(a.bias + n);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
        }
        f32x4 sb = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KD::I8) sb = *(const f32x4*)(a.q.scb + n);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
            const int m = m0 + wr * (WM * 16) + mi * 16 + fr;
            if (m >= a.M) continue;
            const I8Row rw = i8_row<KD>(a, m);
            O4 o;
            float l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) l[j] = gemm_lin<KD, EPI != EPI_BIAS_GELU>(a, acc[ni][mi][j], m, n + j, bv[j], rw, sb[j]);
            if (EPI == EPI_BIAS_RESID) {
                if (rw.defer) {                                   // finished by launch_i8_outlier_side (outlier sum, then the residual)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (OT)l[j];
                    *(O4*)((OT*)a.q.defer_out + (long)blockIdx.z * a.strideC + (long)m * a.ldc + n) = o;
                    continue;
                }
                const O4 rv = *(const O4*)(R + (long)m * a.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (OT)(l[j] + (float)rv[j]);
            } else if (EPI == EPI_BIAS_GELU) {
                if (std::is_same<KD, KBF16>::value && a.gelu_lut) {      // the same table as the 256x256 kernel, read from global memory
                    unsigned t[4]; int idx[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { idx[j] = gelu_lut_index(l[j]); t[j] = a.gelu_lut[gelu_lut_slot(l[j], idx[j])]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_lut_value(l[j], idx[j], t[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (OT)gelu_erf(l[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (OT)l[j];
            }
            *(O4*)(C + (long)m * a.ldc + n) = o;
        }
    }
}

bool gemm256_eligible(const GemmArgs& a, int epi);
void launch_gemm256(const GemmArgs& a, int epi, hipStream_t s);
void launch_gemm256p(const GemmArgs& a, int epi, int cus, hipStream_t s);
thread_local LaunchOpts g_opts;

static void launch_gemm128(const GemmArgs& a, int epi, hipStream_t s);
static int device_cus() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t p; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
    return n;
}
// the 256x256 tile: persistent kernel for the 16-bit kinds (gemm256p.hip), one launch per tile round otherwise (int8 kinds, batched conv stem)
static void launch_tile256(const GemmArgs& a, int epi, hipStream_t s) {
#ifdef SONIC_AB      // the persistent form lost every A/B: it ships only in `make SONIC_AB=1` builds
    if (!a.q.sca && a.batch <= 1 && g_opts.gemm256_persist) { launch_gemm256p(a, epi, (g_opts.gemm256_persist_cus < device_cus() ? g_opts.gemm256_persist_cus : device_cus()), s); return; }
#endif
    launch_gemm256(a, epi, s);
}
// A 256x256 tile owns a CU, so a grid of fewer tiles than CUs leaves the rest of the chip idle for the whole GEMM (one 5 s request through
// the encoder: 30 .. 120 tiles per linear on 256 CUs).  The 128x128 kernel cuts the same problem into four times as many tiles, two to a CU,
// at about `gemm_small_eff` % of the big kernel's per-CU rate when both are full: it gets the GEMM whenever its tile rounds, priced that way,
// are fewer.  Both kernels produce the same bits (tests/test_gpu_parity.py test_gemm256_path), so the choice never shows in a result.  Not the
// q|k|v epilogue: its fused RoPE exists only on the 256x256 tile, and rotating before the bf16 rounding is not the separate pass's arithmetic.
static bool small_grid_prefers128(const GemmArgs& a, int epi) {
    const int eff = g_opts.gemm_small_eff;
    if (eff <= 0 || epi == EPI_QKV_VT) return false;
    const long cus = device_cus(), batch = a.batch > 0 ? a.batch : 1;
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256) * batch, t128 = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * batch;
    const long r256 = (t256 + cus - 1) / cus, r128 = (t128 + 2 * cus - 1) / (2 * cus);
    return r128 * 50 < r256 * eff;      // a round of 128x128 tiles is half the work per CU of a round of 256x256 tiles
}
void launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
    if (!g_opts.gemm_force128 && gemm256_eligible(a, epi) && !small_grid_prefers128(a, epi)) {
        // Wave quantisation: a 256x256 tile occupies a whole CU, so (tiles mod CUs) small means a nearly empty extra round (prefill at
        // M = 8320: 33 x 8 = 264 tiles on 256 CUs, two rounds for 1.03).  If cutting the ragged last <= 128 rows off saves a round, those
        // rows go to the 128x128 kernel instead (same math per row; rows are independent in every epilogue but QKV+V^T).
        const int cus = device_cus(), ntn = (a.N + 255) / 256, Mm = (a.M / 256) * 256, tail = a.M - Mm;
        if (tail > 0 && tail <= 128 && Mm >= 512 && epi != EPI_QKV_VT && a.batch <= 1) {
            const long full = (long)((a.M + 255) / 256) * ntn, main_tiles = (long)(Mm / 256) * ntn;
            if ((main_tiles + cus - 1) / cus < (full + cus - 1) / cus) {
                GemmArgs m = a; m.M = Mm;
                launch_tile256(m, epi, s);
                GemmArgs t = a; t.M = tail; t.C = a.C + (long)Mm * a.ldc;
                if (a.R) t.R = a.R + (long)Mm * a.ldr;
                if (a.q.sca) {
                    // int8 GEMM: A is int8 (1 byte per element) and the per-row quantisation data travels with the rows - scales,
                    // unquantised activations of the outlier columns, and the row -> group (request) index.  (Leaving them at row 0
                    // gave the tail rows the scales and outlier lists of the FIRST rows of the batch: the last request of some batch
                    // compositions differed from its solo result, found by tools/find_batch_dependence.py.)
                    t.A = (const bf16_t*)((const int8_t*)a.A + (long)Mm * a.lda);
                    t.q.sca = a.q.sca + Mm; t.q.x16 = a.q.x16 + (long)Mm * a.q.ldx16; t.q.row_off = a.q.row_off + Mm;
                    if (a.q.defer_out) t.q.defer_out = a.q.defer_out + (long)Mm * a.ldc;
                } else {
                    t.A = a.A + (long)Mm * a.lda;
                }
                launch_gemm128(t, epi, s);
                return;
            }
        }
        launch_tile256(a, epi, s);
        return;
    }
    launch_gemm128(a, epi, s);
}
template <typename KD, int EPI, int NS, int WM, int WN> static void launch_gemm128_v(const GemmArgs& a, hipStream_t s) {
    constexpr int TBM = 32 * WM, TBN = 32 * WN;
    const size_t lds = (size_t)NS * (TBM + TBN) * BK * 2;
    if (lds > 65536) ensure_dyn_lds((const void*)gemm_kernel<KD, EPI, NS, WM, WN>, (int)lds);
    const int tilesM = (a.M + TBM - 1) / TBM, tilesN = (a.N + TBN - 1) / TBN;
    hipLaunchKernelGGL((gemm_kernel<KD, EPI, NS, WM, WN>), dim3(tilesM * tilesN, 1, a.batch > 0 ? a.batch : 1), dim3(256), lds, s, a);
}
// shape of the launch: 0 = 128 x 128 tiles, two-stage ring, two blocks per CU; 1 = 32 x 64 tiles, four-stage ring (grids of 128 x 128 tiles
// that would leave half of the CUs idle); 2 = 32 x 32 tiles (when even the 32 x 64 grid covers at most half of the CUs; not for the SwiGLU
// epilogue, whose gate / up pairs need 32 columns per wave)
template <typename KD, int EPI> static void launch_gemm128_e(const GemmArgs& a, int shape, hipStream_t s) {
    if constexpr (EPI == EPI_QKV_VT) launch_gemm128_v<KD, EPI, 2, 4, 4>(a, s);
    else if (shape == 2 && EPI != EPI_SWIGLU) launch_gemm128_v<KD, EPI, 4, 1, EPI == EPI_SWIGLU ? 2 : 1>(a, s);
    else if (shape >= 1) launch_gemm128_v<KD, EPI, 4, 1, 2>(a, s);
    else launch_gemm128_v<KD, EPI, 2, 4, 4>(a, s);
}
static void launch_gemm128(const GemmArgs& a, int epi, hipStream_t s) {
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    const int batch = a.batch > 0 ? a.batch : 1;
    int shape = (!g_opts.gemm128_shallow && (long)tilesM * tilesN * batch <= device_cus() / 2 && a.K / (128 / (a.q.sca ? 1 : 2)) >= 4) ? 1 : 0;
    if (shape == 1 && (long)((a.M + 31) / 32) * ((a.N + 63) / 64) * batch <= device_cus() / 2) shape = 2;
    KD_SWITCH(a, KD, {
        switch (epi) {
            case EPI_BIAS: launch_gemm128_e<KD, EPI_BIAS>(a, shape, s); break;
            case EPI_BIAS_GELU: launch_gemm128_e<KD, EPI_BIAS_GELU>(a, shape, s); break;
            case EPI_BIAS_RESID: launch_gemm128_e<KD, EPI_BIAS_RESID>(a, shape, s); break;
            case EPI_SWIGLU: launch_gemm128_e<KD, EPI_SWIGLU>(a, shape, s); break;
            case EPI_QKV_VT: if constexpr (!KD::I8) launch_gemm128_e<KD, EPI_QKV_VT>(a, shape, s); break;
        }
    });
}

// ------------------------------------------------------------------------------------------------
// skinny_kernel: decode-step GEMM, M <= 64 rows.  HBM-bound weight streaming: every weight byte is read once, straight
// to VGPRs (no LDS round trip for an operand no other wave shares).  The step is latency-bound, not bandwidth-bound, so
// the kernel is "one-shot": a block owns 16 weight rows x (8 waves * KW * 32) of K; every wave issues ALL of its weight
// loads (nontemporal, 1 KiB each) and activation loads before its first MFMA, so the whole matrix is in flight at once.
// D[n][m] (A-operand = W rows, B-operand = X rows); the 8 K-slices of a block are summed through LDS in fixed order
// (deterministic, no float atomics) and stored as fp32.  K = 2048 needs no split at all (one slab); down_proj
// (K = 6144) leaves 3 slabs for its consumer.
template <typename KD, int MB, int KW, bool NT>
__global__ __launch_bounds__(512) void skinny_kernel(SkinnyArgs a) {
    typedef typename KD::elem ET_; typedef typename KD::frag Frag;
    __shared__ f32x4 red[8][MB][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kb = (blockIdx.y * 8 + wid) * (KW * 32);
    // fragment-tiled weights (tile_weights_kernel): k-step s of row tile t is the contiguous 1 KiB block (t*K/32 + s)
    const ET_* wp = (const ET_*)a.W + ((long)blockIdx.x * (a.K >> 5) + (kb >> 5)) * 512 + lane * 8;
    Frag wf[KW];
#pragma unroll
    for (int u = 0; u < KW; ++u) wf[u] = NT ? __builtin_nontemporal_load((const Frag*)(wp + u * 512)) : *(const Frag*)(wp + u * 512);
    Frag xf[MB][KW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int row = mb * 16 + r; row = row < a.M ? row : a.M - 1;
        const ET_* xp = (const ET_*)a.X + (long)row * a.ldx + kb + g * 8;
#pragma unroll
        for (int u = 0; u < KW; ++u) xf[mb][u] = *(const Frag*)(xp + u * 32);
    }
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KW; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = KD::mfma(wf[u], xf[mb][u], acc[mb]);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) red[wid][mb][lane] = acc[mb];
    __syncthreads();
    const int mpad = MB * 16;
    for (int o = tid; o < 16 * mpad; o += 512) {
        const int m = o >> 4, nl = o & 15, mb = m >> 4, ln = (nl >> 2) * 16 + (m & 15), j = nl & 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += red[w][mb][ln][j];
        a.P[((long)blockIdx.y * mpad + m) * a.N + n0 + nl] = s;
    }
}

// floor: read the same weight bytes, fully coalesced 16 B/lane, one-shot, trivially reduced (bench only)
template <int KW>
__global__ __launch_bounds__(512) void skinny_readfloor_kernel(SkinnyArgs a) {
    const int tid = threadIdx.x;
    const bf16_t* base = a.W + ((long)blockIdx.x * gridDim.y + blockIdx.y) * (8L * KW * 512) ;
    bf16x8 v[KW];
#pragma unroll
    for (int u = 0; u < KW; ++u) v[u] = __builtin_nontemporal_load((const bf16x8*)(base + ((long)u * 512 + tid) * 8));
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < KW; ++u) s += bf2f(v[u][0]) + bf2f(v[u][7]);
    if (s == 123.456f) a.P[0] = s;
}

// skinny_xs_kernel: the decode-step GEMM with the activation slice SHARED through LDS.  Measured on MI355X (tools/
// bench_skinny.py): the weight stream is not the limiter of the one-shot kernels above -- their time scales with M,
// i.e. with the per-wave 64-byte-per-row gathers of X out of L2.  Here a block owns BN = 16*WN weight rows and a K slice
// of BKk = WK*KSW*32; the X slice [M][BKk] is DMA'd ONCE per block into LDS in full 128-byte lines (swizzled on the
// source address, conflict-free ds_read_b128 fragments) and read by all 8 waves; weights go straight to VGPRs from the
// fragment-tiled copy (every wave load is one contiguous 1 KiB, each weight byte read once, nontemporal).  The WK
// K-slices of a block are summed through LDS; K/BKk slabs are left for the consumer (2 for K = 2048 with BKk = 1024).
// NT = 16-row weight tiles per wave (default 1).  The per-CU vector-memory pipe bounds these kernels (W bytes + the X image of every
// block that lands on the CU), so at 48-64 activation rows - where the image of a 1024-deep slice is 64 KiB of int8 - a block should own as
// many weight rows as keeps the grid at one block per CU: gate/up of the full-size model as 96 rows x 1024 (128 x 2 = 256 blocks, 96 KiB
// of W per 64 KiB image) instead of 32 rows x 1024 (768 blocks, three images per CU).
// XQ (int8 kind): the activation rows arrive unquantised (fp16) with their absmax; the block quantises its slice on the way into LDS.  This
// removes the separate one-block-per-row quantisation launch between a producer that does not own whole rows (decode attention: one block
// per (row, kv head); the fused gate/up kernel: 24 columns per block) and the projection that consumes it.
// PRE (16-bit kinds, M <= 2; round 6): the block computes its X slice itself from the PREVIOUS projection's slabs - X = RMSNorm(x + sum of slabs), the
// arithmetic of add_rmsnorm_kernel statement by statement (thread c of a row's 256 owns columns 8c .. 8c + 7; slabs added in ascending order from 0.f;
// sum of squares per thread in column order, wave butterfly, the row's four wave partials in order) - so the standalone add+RMSNorm launch between
// down_proj and the next q|k|v projection (or the lm_head) disappears at no change of any bit.  Every block redoes the whole row (it needs the row's
// sum of squares): 68 KiB of slab and residual reads per row and block out of L2, which pays below three rows.  The updated residual row is written
// by block (0, 0) to a SECOND buffer (other blocks still read the old one), so the residual stream ping-pongs between two buffers layer by layer.
template <typename KD, int MB, int WN, int WK, int KSW, int NT = 1, bool XQ = false, bool PRE = false>
__global__ __launch_bounds__(512) void skinny_xs_kernel(SkinnyArgs a) {
    typedef typename KD::elem ET_; typedef typename KD::frag Frag; typedef typename KD::acc Acc;
    // element size; elements per 16-B chunk, per MFMA k-step, per 128-B LDS row, per 1-KiB weight tile
    constexpr int EB = sizeof(ET_), CE = 16 / EB, KS = 64 / EB, ROWE = 128 / EB, TILE_E = 1024 / EB;
    constexpr int BKk = WK * KSW * KS, NKB = BKk / ROWE, RG = MB * 2, NI = NKB * RG, KBS = MB * 2048, PW = (NI + 7) / 8;
    constexpr int NL = NT * KSW;                                     // weight loads of a wave
    static_assert(WN * WK == 8 && (WK * KSW) % 2 == 0, "8 waves, whole 128-byte K blocks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wn = wid % WN, wk = wid / WN;
    const int n0 = blockIdx.x * (WN * NT * 16) + wn * NT * 16;
    const int kb = blockIdx.y * BKk;
    KT(a, 0);
    // X slice first (small, out of L2): it has to be complete in LDS - for all waves - before the first MFMA
    // XQ: fp16 rows -> registers (asm loads: the compiler must not see them, or its own wait counts would include the weight loads below),
    // G8 = groups of 8 elements per row of the slice; group idx = p * 512 + tid of the MB * 16 * G8 groups: row idx / G8, group idx % G8
    constexpr int G8 = BKk / 8, XTOT = MB * 16 * G8, XP = XQ ? (XTOT + 511) / 512 : 1;
    static_assert(!XQ || KD::I8, "XQ: int8 kind");
    f16x8 xq[XP]; f32x4 xam[XP];
    if constexpr (XQ) {
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int idx = min(p * 512 + tid, XTOT - 1);
            int row = idx / G8; row = row < a.M ? row : a.M - 1;
            const f16_t* src = (const f16_t*)a.X + (long)row * a.ldx + kb + (idx % G8) * 8;
            const float* am = a.x_amax + row * 4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xq[p]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xam[p]) : "v"(am) : "memory");
        }
    } else if constexpr (PRE) {
        // requested below, beside the weights (asm loads + one counted wait: the weights stay in flight while the rows are normalised)
    } else {
        const int lr = lane >> 3, lc = (lane & 7) ^ lr;
#pragma unroll
        for (int t = 0; t < PW; ++t) {
            const int ii = NI % 8 == 0 ? wid * PW + t : wid + t * 8;
            if (NI % 8 == 0 || ii < NI) {
                const int kblock = ii / RG, rg = ii % RG;
                int row = rg * 8 + lr; row = row < a.M ? row : a.M - 1;
                const ET_* src = (const ET_*)a.X + (long)row * a.ldx + kb + kblock * ROWE + lc * CE;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + ii * 1024), 16, 0, 0);
            }
        }
    }
    // PRE: this thread's share of the previous projection's slabs, the residual row and the norm weight (19 loads of 16 bytes), requested first
    static_assert(!PRE || (!KD::I8 && !XQ), "PRE: 16-bit kinds");
    __shared__ float pre_part[8];
    const int prow = tid >> 8, pc = tid & 255;                       // PRE: row 0 -> threads 0 .. 255, row 1 -> 256 .. 511
    const bool pvalid = PRE && prow < a.M && pc < (a.K >> 3);
    f32x4 pnw0, pnw1, psl0[8], psl1[8]; i32x4 pxr;
    if constexpr (PRE) {
        const int cc = pvalid ? pc : 0, rr = prow < a.M ? prow : 0;
        const float* wq = a.pre_w + cc * 8;
        const ET_* xq = (const ET_*)a.pre_x + (long)rr * a.K + cc * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pnw0) : "v"(wq) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(pnw1) : "v"(wq) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pxr) : "v"(xq) : "memory");
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float* sp = a.pre_P + ((long)(ks < a.pre_ks ? ks : 0) * a.pre_mpad + rr) * a.K + cc * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(psl0[ks]) : "v"(sp) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(psl1[ks]) : "v"(sp) : "memory");
        }
    }
    // then the weights (HBM, nontemporal): asm loads with hand-counted waits, so that k-step u is multiplied as soon as ITS fragment
    // has landed (vmcnt retires in order) instead of after the whole slice - the compiler's own bookkeeping falls back to vmcnt(0)
    // when LDS-DMA and register loads are in flight together
    const long tile_stride = (long)(a.K / KS) * TILE_E;              // elements between the 16-row tiles n and n + 16 at one k-step
    const ET_* wp = (const ET_*)a.W + ((long)(n0 >> 4) * (a.K / KS) + ((kb + wk * (KSW * KS)) / KS)) * TILE_E + lane * CE;
    Frag wf[NT][KSW];
#pragma unroll
    for (int u = 0; u < KSW; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const ET_* wpt = wp + t * tile_stride;
            if (u < 4) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(wf[t][u]) : "v"(wpt), "n"(u * 1024) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(wf[t][u]) : "v"(wpt + (u / 4) * 4 * TILE_E), "n"((u % 4) * 1024) : "memory");
        }
    Acc acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][mb][e] = 0;
    KT(a, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");       // this wave's X pieces are in LDS (XQ / PRE: in registers)
    if constexpr (PRE) {
        typedef typename std::conditional<std::is_same<ET_, f16_t>::value, f16x8, bf16x8>::type PV8;
        asm volatile("" : "+v"(pnw0), "+v"(pnw1), "+v"(pxr));
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(psl0[ks]), "+v"(psl1[ks]));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            if (ks < a.pre_ks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] += psl0[ks][j]; acc[4 + j] += psl1[ks][j]; }
            }
        const PV8 tx = __builtin_bit_cast(PV8, pxr);
        PV8 ox; float v[8]; float ssq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ox[j] = (ET_)((float)tx[j] + rT<ET_>(acc[j])); v[j] = (float)ox[j]; ssq += v[j] * v[j]; }
        if (!pvalid) ssq = 0.f;
        if (pvalid && a.pre_xout && blockIdx.x == 0 && blockIdx.y == 0) *(PV8*)((ET_*)a.pre_xout + (long)prow * a.K + pc * 8) = ox;
        ssq = wave_sum(ssq);
        if (lane == 0) pre_part[wid] = ssq;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // (raw barrier: __syncthreads would add a vmcnt(0) fence - the weights are still in flight)
        asm volatile("" ::: "memory");
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) tot += pre_part[(prow & 1) * 4 + i];
        const float rs = 1.0f / sqrtf(tot / a.K + a.pre_eps);
        const int kcol = pc * 8 - kb;                                // this thread's 8 columns inside the block's K slice?
        if (pvalid && kcol >= 0 && kcol < BKk) {
            PV8 oy;
#pragma unroll
            for (int j = 0; j < 8; ++j) oy[j] = (ET_)((j < 4 ? pnw0[j & 3] : pnw1[j & 3]) * rT<ET_>(v[j] * rs));
            const int kblock = kcol / ROWE, ch = (kcol % ROWE) / CE;
            // image rows beyond M hold copies of the last row (as the DMA path's clamped rows): their outputs are never consumed
            for (int m = prow; m < MB * 16; m += (prow == a.M - 1 ? 1 : MB * 16))
                *(PV8*)(smem + kblock * KBS + m * 128 + ((ch ^ (m & 7)) << 4)) = oy;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (XQ) {
        // quantise (elementwise.hip quant_emit_row's arithmetic) and park in the image: byte (row m, element k) of a 128-element k-block at
        // kblock * KBS + m * 128 + ((chunk ^ (m & 7)) << 4) + k % 16
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            asm volatile("" : "+v"(xq[p]), "+v"(xam[p]));
            const int idx = p * 512 + tid;
            if (XTOT % 512 != 0 && idx >= XTOT) continue;
            const int m = idx / G8, g8 = idx % G8, kblock = g8 >> 4, c = (g8 & 15) >> 1, half = g8 & 1;
            const float bm = fmaxf(fmaxf(xam[p][0], xam[p][1]), fmaxf(xam[p][2], xam[p][3])), scale = 127.0f / bm;
            int pk[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = (float)xq[p][j];
                const bool out = !(fabsf(y) < LLM_INT8_THRESHOLD);
                const int qv = (out || !(bm > 0.f)) ? 0 : (int)rintf(y * scale);
                pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
            }
            *(int2*)(smem + kblock * KBS + m * 128 + ((c ^ (m & 7)) << 4) + half * 8) = make_int2(pk[0], pk[1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                    // (raw barrier: __syncthreads would add a vmcnt(0) fence)
    KT(a, 2);
#pragma unroll
    for (int u = 0; u < KSW; ++u) {
        const int kg = wk * KSW + u, kblock = kg >> 1, half = kg & 1;
        Frag xf[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 16 + r;
            xf[mb] = *(const Frag*)(smem + kblock * KBS + m * 128 + (((half * 4 + g) ^ (m & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(wf[t][u]) : "n"(NL - 1 - (u * NT + t)) : "memory");
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[t][mb] = KD::mfma(wf[t][u], xf[mb], acc[t][mb]);
        }
    }
    KT(a, 3);
    __syncthreads();
    KT(a, 4);
    Acc* red = (Acc*)smem;   // [WK][WN * NT][MB][64]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((wk * (WN * NT) + wn * NT + t) * MB + mb) * 64 + lane] = acc[t][mb];
    __syncthreads();
    constexpr int BNR = WN * NT * 16, mpad = MB * 16;
    const int nb0 = blockIdx.x * BNR;
    // One (16-row tile, 16-row block of X) per wave-iteration: a lane adds the WK partials of its own accumulator position (whole fragments,
    // ds_read_b128, ascending k from zero as before - same bits) and stores its four consecutive columns at once.  (Round 4 walked the outputs one
    // by one: WK scalar LDS reads and a 4-byte store each, 4 - 8 store instructions per wave: 1.1 us of a 4.6 us kernel.)
    for (int task = wid; task < WN * NT * MB; task += 8) {
        const int wn2 = task / MB, mb = task % MB;
        Acc sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] = 0;
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            const Acc v = red[((k * (WN * NT) + wn2) * MB + mb) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] += v[e];
        }
        // D[n][m]: lane (r, g) holds columns 4g .. 4g + 3 of X row r; int8: exact int32 partial products, the consumer dequantises (int8_util.h)
        *(Acc*)((typename std::conditional<KD::I8, int, float>::type*)a.P + ((long)blockIdx.y * mpad + mb * 16 + r) * a.N + nb0 + wn2 * 16 + 4 * g) = sum;
    }
    KT(a, 5);
}

// skinny_gu_kernel: decode-step gate/up projection with SwiGLU fused (modeling_llama.py:163-176).
// Weights are fragment-tiled with gate and up rows interleaved in groups of 8 (launch_tile_weights_gu8): one 16-row MFMA tile holds
// gate rows [8t, 8t+8) and up rows [8t, 8t+8) and so yields 8 finished SwiGLU columns.  A block owns TPB consecutive tiles and sees
// the whole K: X [M <= 32][K] sits in LDS once, the 8 waves split K in eighths (each wave: all TPB tiles of its eighth, one X
// fragment read feeds TPB MFMAs), the eighths are summed through LDS in fixed order and the epilogue writes
// act = bf16(bf16(silu(bf16 g)) * bf16 u).  TPB = 3 makes the full-size layer (768 tiles) exactly 256 blocks: the bound of these
// decode kernels is the per-CU vector-memory pipe (~65 GB/s measured for W + the X image), so every CU has to pull its share.
//
// NORM: X is the raw residual stream.  The block turns the o_proj kernel's sum-of-squares partials into the row scale and applies
// RMSNorm (modeling_llama.py:60-65) in place on the staged image, hn = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))); half of the W
// loads are issued up front, the rest go out between the pieces of the normalise pass, so a wave alternates vector-memory issue
// with the VALU work instead of queueing every load first and normalising with the memory pipe idle.
struct GuNorm { const float* SS; int nblk; const float* w; float eps; };   // SS[nblk / 4][32][4] partials, norm weight w

// The K eighths of a block's tiles, summed in fixed order (k = 0 .. 7, from 0.f) and finished: act = bf16(bf16(silu(bf16 g)) * bf16 u).
// red is [wk 8][tile TPB][MB][64 lanes] accumulator fragments; wave w < TPB * MB takes tile j = w / MB, row block mb = w % MB: a lane reads the
// eight fragments of its own position (ds_read_b128, lane-linear), lanes 0-31 then hold four gate columns of one row, lanes 32-63 their up
// partners (D[n][m]: n = 4 * (lane / 16) + e, gate n < 8, up n + 8), one cross-lane move pairs them and lanes 0-31 store four columns at once.
// (Round 3's form walked 768 outputs with 16 scalar LDS reads and a 2-byte store each: 3.8 us of an 11.2 us kernel.)
template <typename T, int MB, int TPB>
__device__ __forceinline__ void gu_reduce_store(const f32x4* red, int wk, int lane, T* act, int ff, int col0, int row0, int m_valid) {
    if (wk < TPB * MB) {                             // wk = task: (tile j, 16-row block mb)
        const int j = wk / MB, mb = wk % MB;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 v = red[((k * TPB + j) * MB + mb) * 64 + lane];
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = __shfl(s[e], (lane + 32) & 63, 64);
        const int m = mb * 16 + (lane & 15);
        if (lane < 32 && m < m_valid) {
            typename ET<T>::v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (T)(rT<T>(silu_f(rT<T>(s[e]))) * rT<T>(u[e]));
            *(typename ET<T>::v4*)(act + (long)(row0 + m) * ff + col0 + j * 8 + (lane >> 4) * 4) = o;
        }
    }
}

// NP = 2 (33 .. 64 rows): the block makes a second pass over rows 32 .. 63 - X image staged into the same LDS, the W fragments are still in
// registers - so W is streamed once for all 64 rows and every row sees exactly the arithmetic of the one-pass kernel (same k order, same
// reduction tree): a request's result does not depend on whether its batch has 8, 32 or 64 rows.
template <typename T, int MB, int KS8, int TPB, bool NORM, int NP = 1>
__global__ __launch_bounds__(512) void skinny_gu_kernel(SkinnyArgs a, T* act, int ff, GuNorm nm) {
    typedef typename ET<T>::v8 V8;
    const T* aX = (const T*)a.X; const T* aW = (const T*)a.W;
    constexpr int K = KS8 * 256, NKB = K / 64, RG = MB * 2, NI = NKB * RG, KBS = MB * 2048, WTOT = TPB * KS8;
    static_assert(NI % 8 == 0, "X pieces must split over 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;   // wave = K eighth
    const int r = lane & 15, g = lane >> 4;
    constexpr int PW = NI / 8, KBW = PW / RG > 0 ? PW / RG : 1;     // X pieces / 64-wide K blocks staged by one wave
    static_assert(PW % RG == 0 || RG % PW == 0, "a wave's pieces must tile whole K blocks or sit inside one");
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    constexpr int SSN = KS8 * 2;                                    // f32x4 of partials per lane: K/16 blocks, two lane halves
    KT(a, 0);
    f32x4 ssv[SSN];
    f32x4 nw[NORM ? KBW * 2 : 1];
    f32x4 nwraw;
    // NORM: scratch behind the X image / the reduction buffer - the 32 row scales (256 B) and 1 KiB of norm weights per wave.  As in skinny_gu64_kernel
    // (see there: the CU's vector-memory path is what bounds these kernels), wave 0 alone fetches the sum-of-squares partials and leaves the row scales
    // in LDS behind a sentinel, and a wave fetches its norm weights with one load.  Round 4: 16 + 8 loads per wave for them, now 2 + 1 on average.
    constexpr int SCR = (KS8 * 4 * MB * 2048 > 8 * TPB * MB * 1024) ? KS8 * 4 * MB * 2048 : 8 * TPB * MB * 1024;
    constexpr unsigned SENT = 0xFFFFFFFFu;
    const unsigned sclds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + SCR, nwlds = sclds + 256 + wk * 1024;
    if (NORM) {
        if (wk == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(sclds + lane * 4), "v"(SENT) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wk == 0) {
            // sum-of-squares partials first: vmcnt retires in order, so they can be consumed while X / W are still in flight.
            // (asm loads + explicit counted waits: with LDS-DMA and plain loads both in flight the compiler's waitcnt pass falls back
            //  to vmcnt(0) at the first use, which would serialise the W fetch behind this pass)
            // SS is [block / 4][32 rows][4]: one load instruction covers 32 rows x 16 B contiguous per lane half (8 cache lines, like a W
            // load); a row-major [row][block] layout would touch 64 lines per instruction and cost as much pipe time as the W fetch.
            const int rl = lane & 31;
            const float* sp = nm.SS + ((long)(lane >> 5) * SSN * 32 + rl) * 4;
#pragma unroll
            for (int i = 0; i < SSN; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ssv[i]) : "v"(sp + (i / 8) * 8 * 128), "n"((i % 8) * 512) : "memory");
        }
    }
#pragma unroll
    for (int t = 0; t < PW; ++t) {                                   // X (NORM: the raw residual) -> LDS by DMA, lane-linear pieces
        const int ii = wk * PW + t, kblock = ii / RG, rg = ii % RG;
        int row = rg * 8 + lr; row = row < a.M ? row : a.M - 1;
        const T* src = aX + (long)row * a.ldx + kblock * 64 + lc * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + ii * 1024), 16, 0, 0);
    }
    // this wave's norm weights: the 64 * KBW floats of its K blocks (KS8 = 1: half a K block per wave, the pair of waves fetches the same 64)
    if (NORM) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nwraw) : "v"(nm.w + (((wk * PW) / RG) * 64 + (lane & (16 * KBW - 1)) * 4)) : "memory");
    constexpr int WPRE = NORM ? WTOT / 2 : WTOT, WREM = WTOT - WPRE;
    V8 wf[WTOT];                                                     // [tile j][k-step u] of this wave's K eighth
    const T* wp = aW + ((long)blockIdx.x * TPB * (K >> 5) + wk * KS8) * 512 + lane * 8;
    // W fragments by inline asm (nt), so that every wait on them is hand-counted: beside LDS-DMA the compiler's own bookkeeping waits
    // vmcnt(0) at the first use of an ordinary load.  The 13-bit instruction offset reaches four 1 KiB fragments per base address.
#define GU_WBASE(f) (wp + ((long)((f) / KS8) * (K >> 5) + (((f) % KS8) / 4) * 4) * 512)
#define GU_WLOAD(f) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(wf[f]) : "v"(GU_WBASE(f)), "n"((((f) % KS8) % 4) * 1024) : "memory")
#pragma unroll
    for (int f = 0; f < WPRE; ++f) GU_WLOAD(f);
    if (NORM && wk == 0) {
        // (behind the wave's other up-front requests: the partials are the oldest, they land while those go out)
        __builtin_amdgcn_sched_barrier(0);
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < SSN; ++i) {
            if (i == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW + 1 + WPRE) : "memory");         // partials landed
            asm volatile("" : "+v"(ssv[i]));                                                         // (uses stay below the wait)
            t += ssv[i][0]; t += ssv[i][1]; t += ssv[i][2]; t += ssv[i][3];
        }
        float t2;
        asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(t2) : "v"((lane ^ 32) << 2), "v"(t) : "memory");
        t = t + t2;
        const float sc = 1.0f / sqrtf(t / (float)K + nm.eps);            // lanes l and l + 32: the scale of row l
        // (a NaN whose bits equal the sentinel - NaN rows in, payload propagated - would read as "not yet": canonical NaN instead)
        asm volatile("ds_write_b32 %0, %1" ::"v"(sclds + (lane & 31) * 4), "v"(__float_as_uint(sc) == SENT ? 0x7FC00000u : __float_as_uint(sc)) : "memory");
    }
    KT(a, 1);
    if (NORM) {
        // Every lane rewrites exactly the 16 bytes its own DMA deposited, so only this wave's vmcnt orders it - no barrier.
        __builtin_amdgcn_sched_barrier(0);            // keep all loads above in flight (the scheduler would sink W below the rewrite)
        // All LDS traffic of this pass is inline asm: the compiler's waitcnt pass treats a DS instruction behind an outstanding
        // LDS-DMA as aliasing it and inserts vmcnt(0), which would also wait for W.
        const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (wk * PW) * 1024 + lane * 16;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPRE) : "memory");             // X pieces and the norm weights landed; W still in flight
        asm volatile("" : "+v"(nwraw));
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(nwlds + lane * 16), "v"(nwraw) : "memory");
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)               // weights of K block kb of this wave, chunk lc: the 8 of this lane's piece column
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(nw[kb * 2]), "=&v"(nw[kb * 2 + 1]) : "v"(nwlds + (kb * 64 + lc * 8) * 4) : "memory");
        {   // the row scales are there when no lane sees the sentinel (wave 0 wrote them ~1 us after entry)
            unsigned sv;
            int spins = 0;                             // (bounded: a wave 0 that never delivers must not hang the GPU; ~1 ms)
            do { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(sv) : "v"(sclds + (lane & 31) * 4) : "memory"); } while (__builtin_amdgcn_ballot_w64(sv == SENT) != 0 && ++spins < (1 << 16));
            // timed out: the sentinel (a NaN) would be used as the row scale - say so where the host looks
            if (spins >= (1 << 16) && lane == 0 && a.err) atomicOr(a.err, 1);
        }
#pragma unroll
        for (int i = 0; i < KBW * 2; ++i) asm volatile("" : "+v"(nw[i]));
#pragma unroll
        for (int tt = 0; tt < PW; ++tt) {
            const int ii = wk * PW + tt, rg = ii % RG, kb = tt / RG;
            int row = rg * 8 + lr; row = row < a.M ? row : a.M - 1;
            float rr; V8 xv;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(rr), "=&v"(xv) : "v"(sclds + row * 4), "v"(lbase + tt * 1024) : "memory");
            const f32x4 w0 = nw[kb * 2], w1 = nw[kb * 2 + 1];
            V8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (T)(w0[j] * rT<T>((float)xv[j] * rr)); o[4 + j] = (T)(w1[j] * rT<T>((float)xv[4 + j] * rr)); }
            asm volatile("ds_write_b128 %0, %1" ::"v"(lbase + tt * 1024), "v"(o) : "memory");
#pragma unroll
            for (int f = WPRE + tt * WREM / PW; f < WPRE + (tt + 1) * WREM / PW; ++f)      // the next W fragment(s) go out behind this piece
                GU_WLOAD(f);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef GU_WLOAD
#undef GU_WBASE
    f32x4 acc[TPB][MB];
#pragma unroll
    for (int j = 0; j < TPB; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[j][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    KT(a, 2);
    // Every wave staged (and, NORM, rewrote) exactly the X pieces of its own K eighth (PW pieces = KS8 / 2 whole 64-wide K blocks), so
    // for KS8 >= 2 nothing here depends on another wave: no barrier before the MFMAs, only this wave's own counted waits.  Fragments
    // were requested in the order f = j * KS8 + u and vmcnt retires in order: tile j is multiplied as soon as ITS fragments have
    // landed, while the later tiles' are still in flight (X, requested before all of W, is covered by the first wait).
    if constexpr (KS8 < 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int f = 0; f < WTOT; ++f) asm volatile("" : "+v"(wf[f]));
        __syncthreads();
    }
    KT(a, 3);
    V8 xfr[KS8][MB];
    if constexpr (KS8 >= 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WTOT - KS8 < 63 ? WTOT - KS8 : 63) : "memory");   // X + tile 0's fragments
        // X fragments by inline asm: behind an LDS-DMA the compiler cannot see retired, it would put vmcnt(0) in front of a ds_read
        const unsigned xbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
        for (int u = 0; u < KS8; ++u) {
            const int kg = wk * KS8 + u, kblock = kg >> 1, half = kg & 1;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mb * 16 + r;
                asm volatile("ds_read_b128 %0, %1" : "=v"(xfr[u][mb]) : "v"(xbase + kblock * KBS + m * 128 + (((half * 4 + g) ^ (m & 7)) << 4)) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < KS8; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) asm volatile("" : "+v"(xfr[u][mb]));
    } else {
#pragma unroll
        for (int u = 0; u < KS8; ++u) {
            const int kg = wk * KS8 + u, kblock = kg >> 1, half = kg & 1;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mb * 16 + r;
                xfr[u][mb] = *(const V8*)(smem + kblock * KBS + m * 128 + (((half * 4 + g) ^ (m & 7)) << 4));
            }
        }
    }
    KT(a, 4);
#pragma unroll
    for (int j = 0; j < TPB; ++j) {
        if constexpr (KS8 >= 2) {
            // fragments of tiles > j may still be in flight: (TPB - 1 - j) * KS8 younger requests
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TPB - 1 - j) * KS8) : "memory");
#pragma unroll
            for (int u = 0; u < KS8; ++u) asm volatile("" : "+v"(wf[j * KS8 + u]));      // (uses stay below the wait)
        }
#pragma unroll
        for (int u = 0; u < KS8; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[j][mb] = ET<T>::mfma(wf[j * KS8 + u], xfr[u][mb], acc[j][mb]);
        __builtin_amdgcn_sched_barrier(0);          // tile j's MFMAs go out before the wait for tile j + 1
    }
    KT(a, 5);
    __syncthreads();
    f32x4* red = (f32x4*)smem;   // [wk 8][tile TPB][MB][64]
#pragma unroll
    for (int j = 0; j < TPB; ++j)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((wk * TPB + j) * MB + mb) * 64 + lane] = acc[j][mb];
    __syncthreads();
    constexpr int NC = 8 * TPB;
    gu_reduce_store<T, MB, TPB>(red, wk, lane, act, ff, blockIdx.x * NC, 0, a.M);
    KT(a, 6);
    if constexpr (NP == 2) {
        // ---- second pass: rows 32 .. 63.  No load is in flight any more (W landed in pass one), so plain waits do.
        const int M2 = a.M - 32;                                        // > 0 by the launcher
        __syncthreads();                                                // the epilogue above is done with the LDS image
        float rl_scale = 0.f;
        if (NORM) {
            const int rl = lane & 31;
            const float* sp = nm.SS + (long)(K / 64) * 128 + ((long)(lane >> 5) * SSN * 32 + rl) * 4;      // region of rows 32 .. 63
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < SSN; ++i) { const f32x4 v = *(const f32x4*)(sp + (long)i * 128); t += v[0]; t += v[1]; t += v[2]; t += v[3]; }
            t = t + __shfl_xor(t, 32, 64);
            rl_scale = 1.0f / sqrtf(t / (float)K + nm.eps);
        }
#pragma unroll
        for (int t = 0; t < PW; ++t) {
            const int ii = wk * PW + t, kblock = ii / RG, rg = ii % RG;
            int row = rg * 8 + lr; row = row < M2 ? row : M2 - 1;
            const T* src = aX + (long)(32 + row) * a.ldx + kblock * 64 + lc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + ii * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NORM) {
            // every lane rewrites the 16 bytes its own DMA deposited (as in pass one); nw still holds this wave's norm weights
#pragma unroll
            for (int tt = 0; tt < PW; ++tt) {
                const int ii = wk * PW + tt, rg = ii % RG, kb = tt / RG;
                int row = rg * 8 + lr; row = row < M2 ? row : M2 - 1;
                const float rr = __shfl(rl_scale, row, 64);
                V8* px = (V8*)(smem + ii * 1024 + lane * 16);
                const V8 xv = *px;
                const f32x4 w0 = nw[kb * 2], w1 = nw[kb * 2 + 1];
                V8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (T)(w0[j] * rT<T>((float)xv[j] * rr)); o[4 + j] = (T)(w1[j] * rT<T>((float)xv[4 + j] * rr)); }
                *px = o;
            }
        }
        if constexpr (KS8 < 2) __syncthreads();                          // (a wave reads other waves' pieces only when its K eighth is half a K block)
        f32x4 acc2[TPB][MB];
#pragma unroll
        for (int j = 0; j < TPB; ++j)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc2[j][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS8; ++u) {
            const int kg = wk * KS8 + u, kblock = kg >> 1, half = kg & 1;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mb * 16 + r;
                const V8 xf = *(const V8*)(smem + kblock * KBS + m * 128 + (((half * 4 + g) ^ (m & 7)) << 4));
#pragma unroll
                for (int j = 0; j < TPB; ++j) acc2[j][mb] = ET<T>::mfma(wf[j * KS8 + u], xf, acc2[j][mb]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TPB; ++j)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) red[((wk * TPB + j) * MB + mb) * 64 + lane] = acc2[j][mb];
        __syncthreads();
        gu_reduce_store<T, MB, TPB>(red, wk, lane, act, ff, blockIdx.x * NC, 32, M2);
        KT(a, 7);
    }
}

// skinny_gu64_kernel: the same op for 33 .. 64 rows in ONE sweep (round 5; round 4's two-pass form is skinny_gu_kernel<.., NP = 2>, kept behind
// sonic_set_option("gu64_two_pass", 1) for the A/B).  The two-pass kernel re-staged rows 32 .. 63 after pass one had finished with LDS: DMA round
// trip + 2.2 us of RMSNorm VALU work + MFMA + a second reduction, 6.4 us in series behind an 10.8 us pass one (profiles/round5_decode_timeline_b64.txt).
// Here the 64 rows go through LDS as FOUR passes of 16 rows over two 64 KiB buffers, and everything a wave does to them depends only on its own
// loads (it stages, normalises and reads the X pieces of its own K eighth - no block barrier before the reduction):
//   pass 0 -> buffer 0, pass 1 -> buffer 1 (up front, beside the partial sums, the norm weights and the first half of W);
//   normalise pass 0, read its fragments into registers, DMA pass 2 into buffer 0;  the same for pass 1 / pass 3 / buffer 1 -
//   so rows 32 .. 63 are staged and normalised WHILE W is still streaming in (W requests are interleaved so that the pass-2 / pass-3 DMA is not
//   behind the last W requests in vmcnt order);  MFMAs of passes 0 / 1 per tile as it lands, then fragments of passes 2 / 3 and their MFMAs;
//   ONE reduction over [8 waves][TPB tiles][4 row blocks] (96 KiB, in the dead buffers).
// Per row the arithmetic is the one-pass kernel's: same MFMA sequence over k, same reduction order, same RMSNorm roundings - a request's bits do
// not depend on the batch size (tests/test_gpu_parity.py::test_fused_decode_rows_vs_unfused, tests/test_gpu_slots.py).
template <typename T, int KS8, int TPB, bool NORM>
__global__ __launch_bounds__(512) void skinny_gu64_kernel(SkinnyArgs a, T* act, int ff, GuNorm nm) {
    typedef typename ET<T>::v8 V8;
    static_assert(KS8 >= 2 && KS8 % 2 == 0, "a wave's K eighth must be whole 64-wide K blocks");
    const T* aX = (const T*)a.X; const T* aW = (const T*)a.W;
    constexpr int K = KS8 * 256, NKB = K / 64, PW = KS8, KBW = KS8 / 2, BUF = NKB * 2048, WTOT = TPB * KS8;
    constexpr int SSN = KS8 * 2;                                    // f32x4 of sum-of-squares partials per lane and 32-row region (as in skinny_gu_kernel)
    // W request schedule (NORM): WPRE up front, WN0 / WN1 between the pieces of normalise passes 0 / 1, WL after the pass-3 DMA
    constexpr int WPRE = NORM ? WTOT / 2 : WTOT, WN0 = (WTOT - WPRE) / 3, WN1 = WN0, WL = WTOT - WPRE - WN0 - WN1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;   // wave = K eighth
    const int r = lane & 15, g = lane >> 4, lr = lane >> 3, lc = (lane & 7) ^ lr;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int mlast = a.M - 1;
    KT(a, 0);
    // What bounds this kernel is the CU's vector-memory path: ~15 ns per wave-wide 16-byte load, whatever it hits (measured with the in-kernel timeline: wave 7
    // got its prologue requests out 4.4 us after wave 0).  Round 4's form had every wave fetch all sum-of-squares partials (2 x 16 KiB) and its 1 KiB of
    // norm weights as eight 8-fold redundant loads: 256 + 64 of the 768 KiB a block pulled.  Now wave 0 alone fetches the partials, adds them in the
    // order of skinny_gu_kernel (same bits) and leaves the 64 row scales in LDS; the other waves find them there (a sentinel cleared behind a barrier at
    // entry marks "not yet": no barrier in the middle of the request stream, which would hold every wave until the last one has issued its prologue);
    // and a wave fetches its norm weights with ONE load and re-reads them in the piece layout through 1 KiB of LDS.
    constexpr int SCR = 2 * BUF > 8 * TPB * 4 * 1024 ? 2 * BUF : 8 * TPB * 4 * 1024;      // scratch behind the buffers / the reduction: 64 scales, 8 x 1 KiB of norm weights
    constexpr unsigned SENT = 0xFFFFFFFFu;
    const unsigned sclds = lds0 + SCR, nwlds = lds0 + SCR + 256 + wk * 1024;
    f32x4 nw[NORM ? KBW * 2 : 1];
    f32x4 nwraw;
    f32x4 ssv[NORM ? SSN : 1], ssw[NORM ? SSN : 1];    // (wave 0)
    if (NORM) {
        if (wk == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(sclds + lane * 4), "v"(SENT) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wk == 0) {
            // partials of region r (rows 32 r ..): lane half h adds groups 16 h .. 16 h + 15 in ascending order, then the halves are added
            const float* ssp = nm.SS + ((long)(lane >> 5) * SSN * 32 + (lane & 31)) * 4;
#pragma unroll
            for (int i = 0; i < SSN; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ssv[i]) : "v"(ssp + (i / 8) * 8 * 128), "n"((i % 8) * 512) : "memory");
#pragma unroll
            for (int i = 0; i < SSN; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ssw[i]) : "v"(ssp + (long)(K / 64) * 128 + (i / 8) * 8 * 128), "n"((i % 8) * 512) : "memory");
        }
    }
    // pass p (rows 16 p ..) of this wave's K eighth -> buffer p & 1, lane-linear 1 KiB pieces [K block][8-row group], chunk ^= row & 7 on the source
#define GU64_STAGE(p) do { const T* xs = aX; asm volatile("" : "+s"(xs));   /* (addresses are computed here, not hoisted into the prologue) */ \
        _Pragma("unroll") for (int t = 0; t < PW; ++t) { \
        const int ii = wk * PW + t, kblock = ii >> 1, rg = ii & 1; \
        int row = (p) * 16 + rg * 8 + lr; row = row < mlast ? row : mlast; \
        const T* src = xs + (long)row * a.ldx + kblock * 64 + lc * 8; \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, \
                                         (__attribute__((address_space(3))) void*)(smem + ((p) & 1) * BUF + ii * 1024), 16, 0, 0); } } while (0)
    GU64_STAGE(0);
    GU64_STAGE(1);
    if (NORM) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nwraw) : "v"(nm.w + (wk * KBW * 64 + lane * 4)) : "memory");    // this wave's 256 norm weights
    V8 wf[WTOT];                                                     // [tile j][k-step u] of this wave's K eighth
    const T* wp = aW + ((long)blockIdx.x * TPB * (K >> 5) + wk * KS8) * 512 + lane * 8;
#define GU_WBASE(f) (wp + ((long)((f) / KS8) * (K >> 5) + (((f) % KS8) / 4) * 4) * 512)
#define GU_WLOAD(f) asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(wf[f]) : "v"(GU_WBASE(f)), "n"((((f) % KS8) % 4) * 1024) : "memory")
#pragma unroll
    for (int f = 0; f < WPRE; ++f) GU_WLOAD(f);
    if (NORM && wk == 0) {                             // (behind its other up-front requests: the partials are the oldest, they land while those go out)
        {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW + 1 + WPRE) : "memory");
            float sc[2];
#define GU64_SS_SUM(reg, sv) do { float t = 0.f; \
            _Pragma("unroll") for (int i = 0; i < SSN; ++i) { asm volatile("" : "+v"(sv[i])); t += sv[i][0]; t += sv[i][1]; t += sv[i][2]; t += sv[i][3]; } \
            float t2; \
            asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(t2) : "v"((lane ^ 32) << 2), "v"(t) : "memory"); \
            t = t + t2; \
            sc[reg] = 1.0f / sqrtf(t / (float)K + nm.eps); } while (0)           /* lane l: the scale of row 32 reg + (l & 31) */
            GU64_SS_SUM(0, ssv);
            GU64_SS_SUM(1, ssw);
#undef GU64_SS_SUM
            const unsigned scb = __float_as_uint(lane < 32 ? sc[0] : sc[1]);            // scale of row `lane`
            asm volatile("ds_write_b32 %0, %1" ::"v"(sclds + lane * 4), "v"(scb == SENT ? 0x7FC00000u : scb) : "memory");   // (a NaN with the sentinel's bits would read as "not yet")
        }
    }
    KTW(a, 1);
    __builtin_amdgcn_sched_barrier(0);                // keep every load above in flight ahead of the VALU work
    // one 16-row pass normalised in place: every lane rewrites exactly the 16 bytes its own DMA deposited (only this wave's vmcnt orders it);
    // hn = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))) (modeling_llama.py:60-65); W requests f0 .. f0 + nf - 1 go out between the pieces
#define GU64_NORM(p, f0, nf) do { _Pragma("unroll") for (int tt = 0; tt < PW; ++tt) { \
        const int rg = tt & 1, kb = tt >> 1; \
        int row = (p) * 16 + rg * 8 + lr; row = row < mlast ? row : mlast; \
        const unsigned pa = lds0 + ((p) & 1) * BUF + (wk * PW + tt) * 1024 + lane * 16; \
        float rr; V8 xv; \
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" \
                     : "=&v"(rr), "=&v"(xv) : "v"(sclds + row * 4), "v"(pa) : "memory"); \
        const f32x4 w0 = nw[kb * 2], w1 = nw[kb * 2 + 1]; \
        V8 o; \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { o[j] = (T)(w0[j] * rT<T>((float)xv[j] * rr)); o[4 + j] = (T)(w1[j] * rT<T>((float)xv[4 + j] * rr)); } \
        asm volatile("ds_write_b128 %0, %1" ::"v"(pa), "v"(o) : "memory"); \
        _Pragma("unroll") for (int f = (f0) + tt * (nf) / PW; f < (f0) + (tt + 1) * (nf) / PW; ++f) GU_WLOAD(f); \
        __builtin_amdgcn_sched_barrier(0); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
    // fragments of one pass: B operand rows = the pass's 16 rows, k = this wave's KS8 steps of 32
#define GU64_FRAGS(p, xf) do { _Pragma("unroll") for (int u = 0; u < KS8; ++u) { \
        const int kg = wk * KS8 + u, kblock = kg >> 1, half = kg & 1; \
        asm volatile("ds_read_b128 %0, %1" : "=v"(xf[u]) : "v"(lds0 + ((p) & 1) * BUF + kblock * 2048 + r * 128 + (((half * 4 + g) ^ (r & 7)) << 4)) : "memory"); } \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        _Pragma("unroll") for (int u = 0; u < KS8; ++u) asm volatile("" : "+v"(xf[u])); } while (0)
    V8 xa[KS8], xb[KS8];                               // passes 0 / 1, later 2 / 3
    if (NORM) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPRE) : "memory");                        // passes 0 / 1 and the norm weights landed; W in flight
        asm volatile("" : "+v"(nwraw));
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(nwlds + lane * 16), "v"(nwraw) : "memory");
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)               // weights of K block kb, chunk lc: the 8 of this lane's piece column
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(nw[kb * 2]), "=&v"(nw[kb * 2 + 1]) : "v"(nwlds + kb * 256 + lc * 32) : "memory");
        {   // the row scales are there when no lane sees the sentinel (wave 0 wrote them ~1 us after entry)
            unsigned sv;
            int spins = 0;                             // (bounded: a wave 0 that never delivers must not hang the GPU; ~1 ms)
            do { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(sv) : "v"(sclds + lane * 4) : "memory"); } while (__builtin_amdgcn_ballot_w64(sv == SENT) != 0 && ++spins < (1 << 16));
            // timed out: the sentinel (a NaN) would be used as the row scale - say so where the host looks
            if (spins >= (1 << 16) && lane == 0 && a.err) atomicOr(a.err, 1);
        }
#pragma unroll
        for (int i = 0; i < KBW * 2; ++i) asm volatile("" : "+v"(nw[i]));
        GU64_NORM(0, WPRE, WN0);
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPRE) : "memory");                        // passes 0 / 1 landed
    }
    GU64_FRAGS(0, xa);
    GU64_STAGE(2);
    if (NORM) GU64_NORM(1, WPRE + WN0, WN1);
    GU64_FRAGS(1, xb);
    GU64_STAGE(3);
#pragma unroll
    for (int f = WTOT - WL; f < WTOT; ++f) GU_WLOAD(f);
    KTW(a, 2);
    // in flight now, in vmcnt order: W[0, WPRE + WN0) | pass 2 (PW) | W[.., + WN1) | pass 3 (PW) | W last WL   (NORM = false: all of W | pass 2 | pass 3)
    f32x4 acc[TPB][4];
#pragma unroll
    for (int j = 0; j < TPB; ++j) { acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // MFMAs of passes 0 / 1 for tiles [j0, j1): requests after tile j1 - 1's last fragment may still be in flight
#define GU64_MMA01(j0, j1) do { _Pragma("unroll") for (int jj = (j0); jj < (j1); ++jj) { \
        _Pragma("unroll") for (int u = 0; u < KS8; ++u) asm volatile("" : "+v"(wf[jj * KS8 + u])); \
        _Pragma("unroll") for (int u = 0; u < KS8; ++u) { \
            acc[jj][0] = ET<T>::mfma(wf[jj * KS8 + u], xa[u], acc[jj][0]); acc[jj][1] = ET<T>::mfma(wf[jj * KS8 + u], xb[u], acc[jj][1]); } } \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    constexpr int JE = NORM ? (WPRE + WN0) / KS8 : 0;                // tiles whose fragments were all requested before pass 2
    if (NORM) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN1 + PW + WL) : "memory");               // pass 2 landed (and tiles < JE)
        GU64_MMA01(0, JE);
        GU64_NORM(2, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WL) : "memory");                          // pass 3 landed
        GU64_NORM(3, 0, 0);
        KTW(a, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KTW(a, 4);
        GU64_MMA01(JE, TPB);
    } else {
#pragma unroll
        for (int j = 0; j < TPB; ++j) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TPB - 1 - j) * KS8 + 2 * PW) : "memory");
            GU64_MMA01(j, j + 1);
        }
        KTW(a, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KTW(a, 4);
    }
    GU64_FRAGS(2, xa);
    GU64_FRAGS(3, xb);
#pragma unroll
    for (int j = 0; j < TPB; ++j) {
        acc[j][2] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[j][3] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS8; ++u) {
            acc[j][2] = ET<T>::mfma(wf[j * KS8 + u], xa[u], acc[j][2]); acc[j][3] = ET<T>::mfma(wf[j * KS8 + u], xb[u], acc[j][3]);
        }
    }
#undef GU64_MMA01
#undef GU64_FRAGS
#undef GU64_NORM
#undef GU64_STAGE
#undef GU_WLOAD
#undef GU_WBASE
    KTW(a, 5);
    __syncthreads();
    f32x4* red = (f32x4*)smem;   // [wk 8][tile TPB][4 row blocks][64]
#pragma unroll
    for (int j = 0; j < TPB; ++j)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) red[((wk * TPB + j) * 4 + mb) * 64 + lane] = acc[j][mb];
    __syncthreads();
    constexpr int NC = 8 * TPB;
#pragma unroll
    for (int task = 0; task < TPB * 4; task += 8) gu_reduce_store<T, 4, TPB>(red, task + wk, lane, act, ff, blockIdx.x * NC, 0, a.M);
    KT(a, 6);
}

// skinny_o_kernel: decode-step o_proj with the residual add fused (modeling_llama.py:306-309).  Like skinny_gu_kernel the block sees the
// whole K (X = attention output, one 16-row group per blockIdx.y, resident in LDS), owns one 16-column tile, sums the 8 K eighths
// through LDS and then writes x = bf16(x + bf16(acc)) in place.  It also emits, per block, the partial sum of squares of its 16
// columns of every updated row (SS[block / 4][row][4], fixed summation order): the consumer (skinny_gu_kernel<NORM>) turns them into the
// RMSNorm scale, so the separate add+RMSNorm kernel between o_proj and gate/up disappears.
template <typename T, int KS8, int MB = 1>
__global__ __launch_bounds__(512) void skinny_o_kernel(SkinnyArgs a, T* x, int ldxres, float* SS) {
    // MB = 2 (33 .. 64 rows): a block takes 32 rows, so the grid stays at one block per CU (128 column tiles x 2) and the 16-column weight tile is
    // fetched once per 32 rows - with 16-row blocks a CU ran two blocks, each with its own W fragments.  Per row the arithmetic is the same.
    typedef typename ET<T>::v8 V8;
    constexpr int K = KS8 * 256, NKB = K / 64, RG = 2 * MB, NI = NKB * RG, KBS = MB * 2048, mpad = 16 * MB;
    static_assert(NI % 8 == 0, "X pieces must split over 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.y * mpad, M = a.M - row0 < mpad ? a.M - row0 : mpad;
    const T* X = (const T*)a.X + (long)row0 * a.ldx;
    x += (long)row0 * ldxres;
    KT(a, 0);
    V8 wf[KS8];
    {
        const T* wp = (const T*)a.W + ((long)blockIdx.x * (K >> 5) + wk * KS8) * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < KS8; ++u) wf[u] = __builtin_nontemporal_load((const V8*)(wp + u * 512));
    }
    {
        const int lr = lane >> 3, lc = (lane & 7) ^ lr;
#pragma unroll
        for (int t = 0; t < NI / 8; ++t) {
            const int ii = wk * (NI / 8) + t, kblock = ii / RG, rg = ii % RG;
            int row = rg * 8 + lr; row = row < M ? row : M - 1;
            const T* src = X + (long)row * a.ldx + kblock * 64 + lc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + ii * 1024), 16, 0, 0);
        }
    }
    // residual value this thread updates in the epilogue (threads 0 .. 256 MB - 1: row tid / 16, column tid % 16), fetched up front
    const int em = (tid >> 4) & (mpad - 1), ec = tid & 15;
    const T xres = x[(long)(em < M ? em : M - 1) * ldxres + blockIdx.x * 16 + ec];
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    KT(a, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KT(a, 2);
    __syncthreads();
    KT(a, 3);
#pragma unroll
    for (int u = 0; u < KS8; ++u) {
        const int kg = wk * KS8 + u, kblock = kg >> 1, half = kg & 1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = mb * 16 + r;
            const V8 xf = *(const V8*)(smem + kblock * KBS + m * 128 + (((half * 4 + g) ^ (m & 7)) << 4));
            acc[mb] = ET<T>::mfma(wf[u], xf, acc[mb]);
        }
    }
    KT(a, 4);
    __syncthreads();
    f32x4* red = (f32x4*)smem;                            // [wk 8][MB][64]
    float* sq = (float*)(smem + 8 * MB * 1024);           // [16 MB rows][16 cols] squares of the updated residual
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) red[(wk * MB + mb) * 64 + lane] = acc[mb];
    __syncthreads();
    if (tid < 256 * MB) {
        const int ln = (ec >> 2) * 16 + (em & 15), j = ec & 3, mb = em >> 4;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[(k * MB + mb) * 64 + ln][j];
        float xn = 0.f;
        if (em < M) {
            xn = rT<T>((float)xres + rT<T>(v));
            x[(long)em * ldxres + blockIdx.x * 16 + ec] = (T)xn;
        }
        sq[em * 16 + ec] = xn * xn;
    }
    __syncthreads();
    if (tid < mpad) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) t += sq[tid * 16 + c];
        // [row0 / 32][block / 4][32 rows][4]: one region per 32 rows (the consumer stages 32 rows per pass), see skinny_gu_kernel
        SS[((long)(row0 >> 5) * (gridDim.x >> 2) * 32 + (long)(blockIdx.x >> 2) * 32 + (row0 & 31) + tid) * 4 + (blockIdx.x & 3)] = t;
    }
    KT(a, 5);
}

template <int MB, int KS8, int TPB, bool NORM, int NP = 1> static void launch_gu_v(const SkinnyArgs& a, bf16_t* act, const GuNorm& nm, hipStream_t s) {
    const size_t lds = (size_t)(KS8 * 4) * MB * 2048;               // X image
    const size_t redb = (size_t)8 * TPB * MB * 1024;
    const size_t need = (lds > redb ? lds : redb) + (NORM ? 256 + 8192 : 0);    // + the row scales and the waves' norm weights
    DT_SWITCH(a.dt, T, {
        if (need > 65536) ensure_dyn_lds((const void*)skinny_gu_kernel<T, MB, KS8, TPB, NORM, NP>, (int)need);
        hipLaunchKernelGGL((skinny_gu_kernel<T, MB, KS8, TPB, NORM, NP>), dim3(a.N / (16 * TPB)), dim3(512), need, s, a, (T*)act, a.N / 2, nm);
    });
}
template <int KS8, int TPB, bool NORM> static void launch_gu64_v(const SkinnyArgs& a, bf16_t* act, const GuNorm& nm, hipStream_t s) {
    const size_t bufs = (size_t)2 * (KS8 * 4) * 2048, redb = (size_t)8 * TPB * 4 * 1024;          // two 16-row X buffers / the reduction over 64 rows
    const size_t need = (bufs > redb ? bufs : redb) + (NORM ? 256 + 8192 : 0);                    // + the row scales and the waves' norm weights
    DT_SWITCH(a.dt, T, {
        if (need > 65536) ensure_dyn_lds((const void*)skinny_gu64_kernel<T, KS8, TPB, NORM>, (int)need);
        SkinnyArgs a2 = a; a2.kt_thread = (g_opts.ktrace_wave & 7) * 64;
        hipLaunchKernelGGL((skinny_gu64_kernel<T, KS8, TPB, NORM>), dim3(a.N / (16 * TPB)), dim3(512), need, s, a2, (T*)act, a.N / 2, nm);
    });
}
// true if the fused kernel handles this shape (else: skinny GEMM + swiglu_slab_kernel)
bool skinny_gu_eligible(int M, int N, int K) {
    if (g_opts.no_fused_gu) return false;
    const int mb = (M + 15) / 16;
    if (mb > 4 || N % 32 || N / 32 < 128 || (mb > 2 && g_opts.no_fused_gu64)) return false;
    return K == 256 || K == 512 || K == 1024 || K == 2048;
}
template <bool NORM> static void launch_gu_any(const SkinnyArgs& a, bf16_t* act, const GuNorm& nm, hipStream_t s) {
    const int mb = (a.M + 15) / 16;
    const bool t3 = (a.N / 16) % 3 == 0;                             // 3 tiles per block where the tile count allows (768 tiles -> 256 blocks)
#define GU(KS8) do { if (mb <= 1) { if (t3) launch_gu_v<1, KS8, 3, NORM>(a, act, nm, s); else launch_gu_v<1, KS8, 2, NORM>(a, act, nm, s); } \
                     else if (mb <= 2) { if (t3) launch_gu_v<2, KS8, 3, NORM>(a, act, nm, s); else launch_gu_v<2, KS8, 2, NORM>(a, act, nm, s); } \
                     else if (KS8 >= 2 && !g_opts.gu64_two_pass) { if (t3) launch_gu64_v<(KS8 >= 2 ? KS8 : 2), 3, NORM>(a, act, nm, s); else launch_gu64_v<(KS8 >= 2 ? KS8 : 2), 2, NORM>(a, act, nm, s); } /* 33 .. 64 rows: one sweep */ \
                     else { if (t3) launch_gu_v<2, KS8, 3, NORM, 2>(a, act, nm, s); else launch_gu_v<2, KS8, 2, NORM, 2>(a, act, nm, s); } } while (0)   /* ... as two passes of 32 (K = 256; A/B) */
    switch (a.K) { case 256: GU(1); break; case 512: GU(2); break; case 1024: GU(4); break; default: GU(8); break; }
#undef GU
}
// W: launch_tile_weights_gu8 layout
void launch_skinny_gu(const SkinnyArgs& a, bf16_t* act, hipStream_t s) { launch_gu_any<false>(a, act, GuNorm{}, s); }
// gate/up on the raw residual: RMSNorm scale from the o_proj kernel's sum-of-squares partials (nblk = K / 16 per row)
void launch_skinny_gu_norm(const SkinnyArgs& a, bf16_t* act, const float* SS, int nblk, const float* w, float eps, hipStream_t s) {
    launch_gu_any<true>(a, act, GuNorm{SS, nblk, w, eps}, s);
}
template <int KS8, int MB> static void launch_o_vm(const SkinnyArgs& a, bf16_t* x, int ldxres, float* SS, hipStream_t s) {
    const size_t lds = (size_t)(KS8 * 4) * MB * 2048, redb = (size_t)MB * (8 * 1024 + 1024);
    const size_t need = lds > redb ? lds : redb;
    DT_SWITCH(a.dt, T, {
        if (need > 65536) ensure_dyn_lds((const void*)skinny_o_kernel<T, KS8, MB>, (int)need);
        hipLaunchKernelGGL((skinny_o_kernel<T, KS8, MB>), dim3(a.N / 16, (a.M + 16 * MB - 1) / (16 * MB)), dim3(512), need, s, a, (T*)x, ldxres, SS);
    });
}
template <int KS8> static void launch_o_v(const SkinnyArgs& a, bf16_t* x, int ldxres, float* SS, hipStream_t s) {
    if (a.M > 32 && !g_opts.o64_16rows) launch_o_vm<KS8, 2>(a, x, ldxres, SS, s); else launch_o_vm<KS8, 1>(a, x, ldxres, SS, s);
}
bool skinny_o_eligible(int M, int N, int K) {
    if (g_opts.no_fused_gu) return false;
    if (M > 64 || N % 16 || (M > 32 && g_opts.no_fused_gu64)) return false;
    return K == 256 || K == 512 || K == 1024 || K == 2048;
}
void launch_skinny_o(const SkinnyArgs& a, bf16_t* x, int ldxres, float* SS, hipStream_t s) {
    switch (a.K) { case 256: launch_o_v<1>(a, x, ldxres, SS, s); break; case 512: launch_o_v<2>(a, x, ldxres, SS, s); break;
                   case 1024: launch_o_v<4>(a, x, ldxres, SS, s); break; default: launch_o_v<8>(a, x, ldxres, SS, s); break; }
}


static int skinny_pick_kw(int K) {
    for (int kw = 8; kw >= 1; kw >>= 1)
        if (K % (256 * kw) == 0 && K / (256 * kw) <= 8) return kw;
    return 0;
}
// config: 1 = BIG (64 rows x 1024 k per block), 3 = 64 rows x 768 k, 2 = SMALL (32 rows x 512 k), 0 = one-shot register kernel (small K)
static int skinny_pick_cfg(int N, int K) {
    if (g_opts.skinny_variant >= 2) return 0;
    // 768-deep slices where they put a block on more CUs than 1024-deep ones without exceeding one block per CU: down_proj of the full-size
    // model (2048 x 6144) is 32 x 8 = 256 blocks instead of 32 x 6 = 192 - the weight stream is bound by how many CUs pull it
    if (!g_opts.no_skinny768 && K % 768 == 0 && N % 64 == 0 && K / 768 <= 8 && (long)(N / 64) * (K / 768) <= 256 &&
        (K % 1024 != 0 || (long)(N / 64) * (K / 768) > (long)(N / 64) * (K / 1024)) && (long)(N / 64) * (K / 768) >= 192) return 3;
    if (K % 1024 == 0 && N % 64 == 0 && K / 1024 <= 8 && (long)(N / 64) * (K / 1024) >= 192) return 1;
    // 48 rows x 512 where that is one block per CU and 32 x 512 is not: the QKV projection of the full-size model (3072 x 2048) as 64 x 4 = 256 blocks
    // instead of 96 x 4 = 384 (half of the CUs got two blocks, each with its own 32 - 64 KiB X image through the CU's vector-memory path)
    if (!g_opts.no_skinny48 && K % 512 == 0 && N % 48 == 0 && K / 512 <= 8 && (long)(N / 48) * (K / 512) <= 256 && (long)(N / 48) * (K / 512) >= 192 &&
        (long)(N / 32) * (K / 512) > 256) return 4;
    if (K % 512 == 0 && N % 32 == 0 && K / 512 <= 8) return 2;
    if (K % 1024 == 0 && N % 64 == 0 && K / 1024 <= 8) return 1;
    return 0;
}
int skinny_pick_ksplit(int N, int K) {
    const int cfg = skinny_pick_cfg(N, K);
    if (cfg == 1) return K / 1024;
    if (cfg == 3) return K / 768;
    if (cfg == 2 || cfg == 4) return K / 512;
    const int kw = skinny_pick_kw(K);
    return kw ? K / (256 * kw) : 0;
}

template <typename KD, int MB, int KW> static void launch_skinny_v(const SkinnyArgs& a, hipStream_t s) {
    dim3 grid(a.N / 16, a.ksplit), block(512);
    const int v = g_opts.skinny_variant;
    if (v == 9) hipLaunchKernelGGL((skinny_readfloor_kernel<KW>), grid, block, 0, s, a);
    else if (v == 3) hipLaunchKernelGGL((skinny_kernel<KD, MB, KW, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((skinny_kernel<KD, MB, KW, true>), grid, block, 0, s, a);
}
template <typename KD, int MB> static void launch_skinny_mb(const SkinnyArgs& a, int kw, hipStream_t s) {
    switch (kw) {
        case 8: launch_skinny_v<KD, MB, 8>(a, s); break;
        case 4: launch_skinny_v<KD, MB, 4>(a, s); break;
        case 2: launch_skinny_v<KD, MB, 2>(a, s); break;
        default: launch_skinny_v<KD, MB, 1>(a, s); break;
    }
}
template <typename KD, int MB, int WN, int WK, int KSW, int NT = 1, bool XQ = false, bool PRE = false> static void launch_xs_v(const SkinnyArgs& a, int nkslices, hipStream_t s) {
    constexpr int EB = sizeof(typename KD::elem), KS = 64 / EB, ROWE = 128 / EB;
    constexpr int NI = (WK * KSW * KS / ROWE) * MB * 2;
    const size_t img = (size_t)NI * 1024, red = (size_t)8 * NT * MB * 1024, lds = img > red ? img : red;
    if (lds > 65536) ensure_dyn_lds((const void*)skinny_xs_kernel<KD, MB, WN, WK, KSW, NT, XQ, PRE>, (int)lds);
    hipLaunchKernelGGL((skinny_xs_kernel<KD, MB, WN, WK, KSW, NT, XQ, PRE>), dim3(a.N / (WN * NT * 16), nkslices), dim3(512), lds, s, a);
}
// PRE form (SkinnyArgs.pre_P): M <= 2, 16-bit kinds, any of the shared-X tilings
bool skinny_pre_eligible(int M, int N, int K) { return M >= 1 && M <= 2 && K % 8 == 0 && K <= 2048 && skinny_pick_cfg(N, K) != 0; }
template <typename KD> static void launch_skinny_xs_pre(const SkinnyArgs& a, int cfg, hipStream_t s) {
    if (cfg == 1) launch_xs_v<KD, 1, 4, 2, 16, 1, false, true>(a, a.K / 1024, s);
    else if (cfg == 3) launch_xs_v<KD, 1, 4, 2, 12, 1, false, true>(a, a.K / 768, s);
    else if (cfg == 4) launch_xs_v<KD, 1, 1, 8, 2, 3, false, true>(a, a.K / 512, s);
    else launch_xs_v<KD, 1, 2, 4, 4, 1, false, true>(a, a.K / 512, s);
}
template <typename KD, int MB> static void launch_skinny_xs(const SkinnyArgs& a, int cfg, hipStream_t s) {
    if (cfg == 1) launch_xs_v<KD, MB, 4, 2, 16>(a, a.K / 1024, s);
    else if (cfg == 3) launch_xs_v<KD, MB, 4, 2, 12>(a, a.K / 768, s);
    else if (cfg == 4) launch_xs_v<KD, MB, 1, 8, 2, 3>(a, a.K / 512, s);
    else launch_xs_v<KD, MB, 2, 4, 4>(a, a.K / 512, s);
}
// int8 operands (Linear8bitLt decode step): 32 * NT weight rows x (4 * KSW * 64) of K per block.  The slabs are exact int32 sums, so the
// K split changes no bit of the result and is chosen per shape for one block per CU (full-size model, 256 CUs):
//   cfg 1  96 rows x 1024   gate/up 128 x 2 = 256 blocks          (N % 96 == 0, K % 1024 == 0, >= 192 blocks)
//   cfg 2  64 rows x  768   down     32 x 8 = 256 blocks          (N % 64 == 0, K % 768 == 0, <= 8 slices, >= 192 blocks)
//   cfg 3  32 rows x  512   o_proj   64 x 4 = 256 blocks          (K % 512 == 0, <= 8 slices, 32 x 1024 would give < 192 blocks, this <= 320)
//   cfg 0  32 rows x 1024   q/k/v    96 x 2 = 192 blocks          (K % 1024 == 0)
//   cfg 4  32 rows x  256   tiny test configurations              (K % 256 == 0)
static int skinny_i8_cfg(int N, int K) {
    if (N % 32) return -1;
    if (!g_opts.no_skinny_i8_wide) {
        if (K % 1024 == 0 && N % 96 == 0 && (long)(N / 96) * (K / 1024) >= 192) return 1;
        if (K % 768 == 0 && N % 64 == 0 && K / 768 <= 8 && (long)(N / 64) * (K / 768) >= 192 && (long)(N / 64) * (K / 768) <= 320) return 2;
        if (K % 1024 == 0 && (long)(N / 32) * (K / 1024) < 192 && K / 512 <= 8 && (long)(N / 32) * (K / 512) <= 320) return 3;
    }
    if (K % 1024 == 0) return 0;
    if (K % 256 == 0) return 4;
    return -1;
}
int skinny_pick_ksplit_i8(int N, int K) {
    switch (skinny_i8_cfg(N, K)) {
        case 0: case 1: return K / 1024;
        case 2: return K / 768;
        case 3: return K / 512;
        case 4: return K / 256;
        default: return 0;
    }
}
template <int MB> static void launch_skinny_i8(const SkinnyArgs& a, hipStream_t s) {
    if (a.x_amax) {                  // fp16 rows quantised while staged (the configurations the decode step uses for o_proj / down_proj)
        switch (skinny_i8_cfg(a.N, a.K)) {
            case 2: launch_xs_v<KI8, MB, 2, 4, 3, 2, true>(a, a.K / 768, s); break;
            case 3: launch_xs_v<KI8, MB, 2, 4, 2, 1, true>(a, a.K / 512, s); break;
            case 0: case 1: launch_xs_v<KI8, MB, 2, 4, 4, 1, true>(a, a.K / 1024, s); break;
            default: launch_xs_v<KI8, MB, 2, 4, 1, 1, true>(a, a.K / 256, s); break;
        }
        return;
    }
    switch (skinny_i8_cfg(a.N, a.K)) {
        case 1: launch_xs_v<KI8, MB, 2, 4, 4, 3>(a, a.K / 1024, s); break;
        case 2: launch_xs_v<KI8, MB, 2, 4, 3, 2>(a, a.K / 768, s); break;
        case 3: launch_xs_v<KI8, MB, 2, 4, 2>(a, a.K / 512, s); break;
        case 0: launch_xs_v<KI8, MB, 2, 4, 4>(a, a.K / 1024, s); break;
        default: launch_xs_v<KI8, MB, 2, 4, 1>(a, a.K / 256, s); break;
    }
}
template <typename KD> static void launch_skinny_16(const SkinnyArgs& a, hipStream_t s) {
    const int cfg = skinny_pick_cfg(a.N, a.K);
    const int mb = (a.M + 15) / 16;
    if (a.pre_P) { launch_skinny_xs_pre<KD>(a, cfg, s); return; }      // (the caller checked skinny_pre_eligible)
    if (cfg) {
        if (mb <= 1) launch_skinny_xs<KD, 1>(a, cfg, s);
        else if (mb == 2) launch_skinny_xs<KD, 2>(a, cfg, s);
        else if (mb == 3) launch_skinny_xs<KD, 3>(a, cfg, s);
        else launch_skinny_xs<KD, 4>(a, cfg, s);
        return;
    }
    const int kw = skinny_pick_kw(a.K);
    if (mb <= 1) launch_skinny_mb<KD, 1>(a, kw, s);
    else if (mb == 2) launch_skinny_mb<KD, 2>(a, kw, s);
    else if (mb == 3) launch_skinny_mb<KD, 3>(a, kw, s);
    else launch_skinny_mb<KD, 4>(a, kw, s);
}
void launch_skinny(const SkinnyArgs& a, hipStream_t s) {
    if (a.i8) {
        const int mb = (a.M + 15) / 16;
        if (mb <= 1) launch_skinny_i8<1>(a, s);
        else if (mb == 2) launch_skinny_i8<2>(a, s);
        else if (mb == 3) launch_skinny_i8<3>(a, s);
        else launch_skinny_i8<4>(a, s);
        return;
    }
    if (a.dt == DT_F16) launch_skinny_16<KF16>(a, s); else launch_skinny_16<KBF16>(a, s);
}

// W[N][K] row-major -> fragment-tiled: element (n, k) goes to ((n/16)*(K/32) + k/32)*512 + (((k%32)/8)*16 + n%16)*8 + k%8,
// i.e. the 64 lanes of the MFMA A-operand of (row tile, k-step) read 64 consecutive 16-byte pieces.
__global__ void tile_weights_kernel(const bf16_t* w, bf16_t* wt, int N, int K) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-element piece per thread
    if (e >= (long)N * (K >> 3)) return;
    const int n = e / (K >> 3), kc = e % (K >> 3), k = kc * 8;
    const long dst = ((long)(n >> 4) * (K >> 5) + (k >> 5)) * 512 + ((((k & 31) >> 3) * 16) + (n & 15)) * 8;
    *(bf16x8*)(wt + dst) = *(const bf16x8*)(w + (long)n * K + k);
}
// gate/up variant: the source has gate and up rows interleaved in groups of 16 (the prefill GEMM's SwiGLU epilogue layout); the tiled
// copy interleaves them in groups of 8, so tile t = gate rows [8t, 8t+8) followed by up rows [8t, 8t+8)
__global__ void tile_weights_gu8_kernel(const bf16_t* w, bf16_t* wt, int N, int K) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * (K >> 3)) return;
    const int n = e / (K >> 3), kc = e % (K >> 3), k = kc * 8;       // n = destination row
    const int t = n >> 4, i = n & 15, q = t * 8 + (i & 7);           // q = gate/up row index
    const int src = (q >> 4) * 32 + (i >> 3) * 16 + (q & 15);
    const long dst = ((long)t * (K >> 5) + (k >> 5)) * 512 + ((((k & 31) >> 3) * 16) + i) * 8;
    *(bf16x8*)(wt + dst) = *(const bf16x8*)(w + (long)src * K + k);
}
void launch_tile_weights_gu8(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t s) {
    const long n = (long)N * (K >> 3);
    hipLaunchKernelGGL(tile_weights_gu8_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wt, N, K);
}
void launch_tile_weights(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t s) {
    const long n = (long)N * (K >> 3);
    hipLaunchKernelGGL(tile_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wt, N, K);
}

// int8 variant: a (16-row, 64-k) MFMA A-operand tile of v_mfma_i32_16x16x64_i8 is 1 KiB; element (n, k) goes to
// ((n/16)*(K/64) + k/64)*1024 + (((k%64)/16)*16 + n%16)*16 + k%16
__global__ void tile_weights_i8_kernel(const int8_t* w, int8_t* wt, int N, int K) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece per thread
    if (e >= (long)N * (K >> 4)) return;
    const int n = e / (K >> 4), kc = e % (K >> 4), k = kc * 16;
    const long dst = ((long)(n >> 4) * (K >> 6) + (k >> 6)) * 1024 + ((((k & 63) >> 4) * 16) + (n & 15)) * 16;
    *(i32x4*)(wt + dst) = *(const i32x4*)(w + (long)n * K + k);
}
void launch_tile_weights_i8(const int8_t* w, int8_t* wt, int N, int K, hipStream_t s) {
    const long n = (long)N * (K >> 4);
    hipLaunchKernelGGL(tile_weights_i8_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wt, N, K);
}
