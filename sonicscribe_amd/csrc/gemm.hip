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
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES ((BM + BN) * BK * 2)

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    // ---- tile id: XCD-aware (blocks b, b+8, ... share an L2) + grouped raster (8 tile-rows per group)
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
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
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* A = a.A + (long)blockIdx.z * a.strideA;
    bf16_t* C = a.C + (long)blockIdx.z * a.strideC;
    const bf16_t* R = (EPI == EPI_BIAS_RESID) ? a.R + (long)blockIdx.z * a.strideR : nullptr;

    // ---- per-lane DMA source pointers (4 row groups of 8 rows per wave, per operand)
    const int lr = lane >> 3, lp = lane & 7, lc = lp ^ lr;  // LDS row-in-group, physical chunk, logical chunk
    const bf16_t* srcA[4];
    const bf16_t* srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wid * 4 + i;
        int ra = m0 + g * 8 + lr; ra = ra < a.M ? ra : a.M - 1;
        int rw = n0 + g * 8 + lr; rw = rw < a.N ? rw : a.N - 1;
        srcA[i] = A + (long)ra * a.lda + lc * 8;
        srcW[i] = a.W + (long)rw * a.K + lc * 8;
    }
    auto stage_load = [&](int stage, int k0) {
        char* sA = smem + stage * STAGE_BYTES;
        char* sB = sA + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wid * 4 + i;
            glds16(srcA[i] + k0, sA + g * 1024);
            glds16(srcW[i] + k0, sB + g * 1024);
        }
    };

    f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool vtile = (EPI == EPI_QKV_VT) && (n0 >= a.n_split);
    const int fr = lane & 15, fg = lane >> 4;
    const int nk = a.K / BK;
    stage_load(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage_load(cur ^ 1, (kt + 1) * BK);
        const char* sA = smem + cur * STAGE_BYTES;
        const char* sB = sA + BM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rx = wr * 64 + i * 16 + fr, rw = wc * 64 + i * 16 + fr;
                const int c = kk * 4 + fg;
                xf[i] = *(const bf16x8*)(sA + rx * 128 + ((c ^ (rx & 7)) << 4));
                wf[i] = *(const bf16x8*)(sB + rw * 128 + ((c ^ (rw & 7)) << 4));
            }
            if (vtile) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mi], wf[ni], acc[ni][mi], 0, 0, 0);
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
    if (vtile) {
        // acc[ni][mi][j] = D[m = mrow + j][n = ncol]; V^T[seg][n - n_split][t .. t+3]
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc * 64 + ni * 16 + fr;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0 + wr * 64 + mi * 16 + fg * 4;
                if (m < a.M && n < a.N) {
                    const int seg = m / a.seg_T, t = m - seg * a.seg_T;
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[ni][mi][j] + bv);
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
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0 + wr * 64 + mi * 16 + fr;
                if (m < a.M && (n0 + wc * 64 + q * 32) < a.N) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = rbf(acc[2 * q][mi][j]), u = rbf(acc[2 * q + 1][mi][j]);
                        o[j] = f2bf(rbf(silu_f(g)) * u);
                    }
                    *(bf16x4*)(C + (long)m * a.ldc + oc) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc * 64 + ni * 16 + fg * 4;
        if (n >= a.N) continue;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const f32x4 b4 = *(const f32x4*)(a.bias + n);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr * 64 + mi * 16 + fr;
            if (m >= a.M) continue;
            bf16x4 o;
            if (EPI == EPI_BIAS_RESID) {
                const bf16x4 rv = *(const bf16x4*)(R + (long)m * a.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(rbf(acc[ni][mi][j] + bv[j]) + bf2f(rv[j]));
            } else if (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(gelu_erf(rbf(acc[ni][mi][j] + bv[j])));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(acc[ni][mi][j] + bv[j]);
            }
            *(bf16x4*)(C + (long)m * a.ldc + n) = o;
        }
    }
}

void launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    dim3 grid(tilesM * tilesN, 1, a.batch > 0 ? a.batch : 1), block(256);
    const size_t lds = 2 * STAGE_BYTES;
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL(gemm_kernel<EPI_BIAS>, grid, block, lds, s, a); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_kernel<EPI_BIAS_GELU>, grid, block, lds, s, a); break;
        case EPI_BIAS_RESID: hipLaunchKernelGGL(gemm_kernel<EPI_BIAS_RESID>, grid, block, lds, s, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL(gemm_kernel<EPI_SWIGLU>, grid, block, lds, s, a); break;
        case EPI_QKV_VT: hipLaunchKernelGGL(gemm_kernel<EPI_QKV_VT>, grid, block, lds, s, a); break;
    }
}

// ------------------------------------------------------------------------------------------------
// skinny_kernel: decode-step GEMM, M <= 64 rows.  HBM-bound weight streaming: every weight byte is
// read once, straight to VGPRs (no LDS round trip for an operand no other wave shares).  grid =
// (N/64, ksplit); wave w of a block owns weight rows n0 + 16w .. +15 over the block's K slice and
// emits D[n][m] (A-operand = W rows, B-operand = X rows) so its fp32 partials store as 16-byte
// quads.  Partials land in per-slice slabs summed in fixed order by the consumer kernel
// (deterministic; no float atomics).
template <int MB>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 64 + wid * 16;
    const int kslice = a.K / a.ksplit, kb = blockIdx.y * kslice;
    const bf16_t* wp = a.W + (long)(n0 + r) * a.K + kb + g * 8;
    const bf16_t* xp[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int row = mb * 16 + r; row = row < a.M ? row : a.M - 1;
        xp[mb] = a.X + (long)row * a.ldx + kb + g * 8;
    }
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K slice is a multiple of 128: chunks of 4 k-steps, next chunk's weight loads issued before
    // the current chunk's MFMAs (register double buffer) so >= 4 KiB per wave stays in flight.
    bf16x8 wf[4], wn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wf[u] = __builtin_nontemporal_load((const bf16x8*)(wp + u * 32));
    for (int k = 0; k < kslice; k += 128) {
        const bool more = (k + 128) < kslice;
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) wn[u] = __builtin_nontemporal_load((const bf16x8*)(wp + k + 128 + u * 32));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bf16x8 xf = *(const bf16x8*)(xp[mb] + k + u * 32);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf, acc[mb], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) wf[u] = wn[u];
        }
    }
    const int mpad = MB * 16;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        float* p = a.P + ((long)blockIdx.y * mpad + mb * 16 + r) * a.N + n0 + g * 4;
        *(f32x4*)p = acc[mb];
    }
}

int skinny_pick_ksplit(int N, int K) {
    const int tiles = N / 64;
    int ks = 1;
    while (ks < 8 && tiles * ks < 256 && (K / (ks * 2)) % 32 == 0 && K / (ks * 2) >= 128) ks *= 2;
    return ks;
}

void launch_skinny(const SkinnyArgs& a, hipStream_t s) {
    dim3 grid(a.N / 64, a.ksplit), block(256);
    const int mb = (a.M + 15) / 16;
    if (mb <= 1) hipLaunchKernelGGL(skinny_kernel<1>, grid, block, 0, s, a);
    else if (mb == 2) hipLaunchKernelGGL(skinny_kernel<2>, grid, block, 0, s, a);
    else if (mb == 3) hipLaunchKernelGGL(skinny_kernel<3>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(skinny_kernel<4>, grid, block, 0, s, a);
}
