// LLM.int8 quantisers (spec: oracle/sonic_oracle.c quantize_rows_int8 / linear_int8, which restate bitsandbytes'
// Int8Params.cuda() and functional.int8_vectorwise_quant(A, threshold=6.0) as called from backend/asr.py:182-198).
//
//   weights      CB[n][k] = rn(W[n][k] * 127 / SCB[n]),  SCB[n] = max_k |W[n][k]|                      (once, at sonic_finalize_weights)
//   activations  one Linear8bitLt call sees the rows of one request ("group"): a column is an outlier column of the group if any
//                of its rows holds |x| >= 6 there; SCA[m] = max_k |x[m][k]| over the elements below the threshold;
//                CA[m][k] = rn(x * 127 / SCA[m]), 0 for elements >= 6 and for whole outlier columns; the outlier columns are listed
//                in ascending order for the fp16 side product of the GEMM epilogue.
// HBM-bound streaming passes: 16 B per lane in, 8 B per lane out.  The decode step (every row its own group) uses the one-block-per-row
// producers in elementwise.hip instead.
#include "common.h"
#include "kernels.h"
#include "int8_util.h"

// ---------------------------------------------------------------- weights: one block per row
__global__ __launch_bounds__(256) void quant_weights_kernel(const f16_t* w, int8_t* cb, float* scb, int K) {
    __shared__ float part[4];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const f16_t* r = w + (long)n * K;
    float amax = -1.17549435e-38f;
    for (int c = tid; c < (K >> 3); c += 256) {
        const f16x8 t = *(const f16x8*)(r + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)t[j]));
    }
    amax = wave_max(amax);
    if (lane == 0) part[wid] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (tid == 0) scb[n] = amax;
    const float scale = 127.0f / amax;
    for (int c = tid; c < (K >> 3); c += 256) {
        const f16x8 t = *(const f16x8*)(r + c * 8);
        int pk[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qv = amax > 0.f ? (int)rintf((float)t[j] * scale) : 0;
            pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
        }
        *(int2*)(cb + (long)n * K + c * 8) = make_int2(pk[0], pk[1]);
    }
}
void launch_quant_weights(const bf16_t* w, int8_t* cb, float* scb, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(quant_weights_kernel, dim3(N), dim3(256), 0, s, (const f16_t*)w, cb, scb, K);
}

// ---------------------------------------------------------------- activations, pass 1: row absmax + outlier flags of the group
// one wave per row
__global__ __launch_bounds__(256) void qa_stats_kernel(QuantActArgs a) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int g = a.gmap ? a.gmap[m / a.gdiv] : m / a.gdiv;
    const f16_t* xr = (const f16_t*)a.X + (long)m * a.ld;
    unsigned char* fl = a.flags + (long)g * a.K;
    float amax = -1.17549435e-38f;
    for (int c = lane; c < (a.K >> 3); c += 64) {
        const f16x8 t = *(const f16x8*)(xr + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = fabsf((float)t[j]);
            if (v < LLM_INT8_THRESHOLD) amax = fmaxf(amax, v);
            else fl[c * 8 + j] = 1;                       // (benign race: every writer stores the same value)
        }
    }
    amax = wave_max(amax);
    if (lane == 0) a.sca[m] = amax;
}
// pass 2: int8 rows
__global__ __launch_bounds__(256) void qa_quant_kernel(QuantActArgs a) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int g = a.gmap ? a.gmap[m / a.gdiv] : m / a.gdiv;
    const f16_t* xr = (const f16_t*)a.X + (long)m * a.ld;
    const unsigned char* fl = a.flags + (long)g * a.K;
    const float amax = a.sca[m];
    const float scale = 127.0f / amax;
    for (int c = lane; c < (a.K >> 3); c += 64) {
        const f16x8 t = *(const f16x8*)(xr + c * 8);
        const unsigned long long f8 = *(const unsigned long long*)(fl + c * 8);
        int pk[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (float)t[j];
            const bool zero = ((f8 >> (8 * j)) & 0xFF) || !(fabsf(v) < LLM_INT8_THRESHOLD) || !(amax > 0.f);
            const int qv = zero ? 0 : (int)rintf(v * scale);
            pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
        }
        *(int2*)(a.q + (long)m * a.K + c * 8) = make_int2(pk[0], pk[1]);
    }
}
// pass 3: ascending list of the outlier columns of every group; one block per group
__global__ __launch_bounds__(256) void qa_lists_kernel(QuantActArgs a) {
    __shared__ int cnts[256];
    const int g = blockIdx.x, tid = threadIdx.x;
    const unsigned char* fl = a.flags + (long)g * a.K;
    const int per = (a.K + 255) / 256, k0 = tid * per, k1 = min(a.K, k0 + per);
    int c = 0;
    for (int k = k0; k < k1; ++k) c += fl[k] != 0;
    cnts[tid] = c;
    __syncthreads();
    int base = 0, total = 0;
    for (int t = 0; t < 256; ++t) { if (t < tid) base += cnts[t]; total += cnts[t]; }
    if (tid == 0) a.oc_cnt[g] = total;
    for (int k = k0; k < k1; ++k) if (fl[k]) a.oc_list[(long)g * a.oc_ld + base++] = k;
}
// after qa_lists when the producer quantised the rows before the group's flags were complete: a column some OTHER row of the group flagged
// still holds this row's code; one thread per row walks its group's list (empty for most groups)
__global__ __launch_bounds__(256) void qa_fix_kernel(QuantActArgs a) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= a.M) return;
    const int g = a.gmap ? a.gmap[m / a.gdiv] : m / a.gdiv;
    const int cnt = a.oc_cnt[g];
    for (int i = 0; i < cnt; ++i) a.q[(long)m * a.K + a.oc_list[(long)g * a.oc_ld + i]] = 0;
}
void launch_quant_act_begin(const QuantActArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    launch_fill_i32((int*)a.flags, 0, (int)(((long)a.G * a.K + 3) / 4), s);
}
void launch_quant_act_finish(const QuantActArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    hipLaunchKernelGGL(qa_lists_kernel, dim3(a.G), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qa_fix_kernel, dim3((a.M + 255) / 256), dim3(256), 0, s, a);
}
// passes 1 + 2 in one: a row's absmax does not depend on the other rows of its group (only its OWN elements >= 6.0 are left out), so the wave
// that read the row can quantise it at once - own outliers as 0 - and raise their flags; the columns OTHER rows flag are zeroed afterwards
// by qa_fix, as for the rows a LayerNorm quantised.  One read of X instead of two.  NCH * 512 >= K.
template <int NCH>
__global__ __launch_bounds__(256) void qa_rowquant_kernel(QuantActArgs a) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int g = a.gmap ? a.gmap[m / a.gdiv] : m / a.gdiv;
    const f16_t* xr = (const f16_t*)a.X + (long)m * a.ld;
    unsigned char* fl = a.flags + (long)g * a.K;
    const int nv = a.K >> 3;
    f16x8 t[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { const int c = lane + i * 64; if (c < nv) t[i] = *(const f16x8*)(xr + c * 8); }
    float amax = -1.17549435e-38f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + i * 64 < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float v = fabsf((float)t[i][j]); if (v < LLM_INT8_THRESHOLD) amax = fmaxf(amax, v); }
        }
    amax = wave_max(amax);
    if (lane == 0) a.sca[m] = amax;
    const float scale = 127.0f / amax;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            int pk[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = (float)t[i][j];
                const bool out = !(fabsf(v) < LLM_INT8_THRESHOLD);
                if (out) fl[c * 8 + j] = 1;                       // (benign race: every writer stores the same value)
                const int qv = (out || !(amax > 0.f)) ? 0 : (int)rintf(v * scale);
                pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
            }
            *(int2*)(a.q + (long)m * a.K + c * 8) = make_int2(pk[0], pk[1]);
        }
    }
}
void launch_quant_act(const QuantActArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    if (!getenv("SONIC_QA_3PASS") && a.K % 8 == 0 && a.K <= 12 * 512) {
        launch_fill_i32((int*)a.flags, 0, (int)(((long)a.G * a.K + 3) / 4), s);
        if (a.K <= 4 * 512) hipLaunchKernelGGL(qa_rowquant_kernel<4>, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(qa_rowquant_kernel<12>, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
        hipLaunchKernelGGL(qa_lists_kernel, dim3(a.G), dim3(256), 0, s, a);
        hipLaunchKernelGGL(qa_fix_kernel, dim3((a.M + 255) / 256), dim3(256), 0, s, a);
        return;
    }
    launch_fill_i32((int*)a.flags, 0, (int)(((long)a.G * a.K + 3) / 4), s);
    hipLaunchKernelGGL(qa_stats_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qa_quant_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qa_lists_kernel, dim3(a.G), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- deferred outlier columns of an int8 GEMM (prefill / encoder, RESID epilogue)
// bitsandbytes adds the outlier columns of a Linear8bitLt call as a dense fp16 matmul over the gathered columns
// (MatMul8bitLt.forward: output.addmm(subA, subB)); the GEMM epilogues walk the list per output element instead, which is the right
// trade for the handful of columns a real checkpoint has but costs O(columns) scattered loads per element: 1.1 ms per down_proj GEMM for
// the ~360 columns synthetic SwiGLU activations produce (profiles/round2_int8_b64_kernel_summary.txt).  Rows whose group lists more
// than defer_thr columns therefore leave the GEMM as v = fp16(acc * s + b) in `tmp`, and this kernel finishes them:
//     out = fp16(fp16(v + sum_k x[m][k] * wdq[n][k]) + R[m][n]),  wdq = fp16(CB[n][k] * SCB[n] / 127)   (oracle/sonic_oracle.c linear_int8)
// with the sum on the matrix pipe: v_mfma_f32_16x16x32_f16 over 32-column chunks of the group's ascending list, operands gathered into
// LDS (products exact, fp32 accumulation in the MFMA's own order: the scalar epilogue sums in ascending k, so a deferred element can
// differ from it by one fp16 ulp when v + sum lands on a rounding boundary; the order depends on the list only, never on the batch).
// Block: 64 rows x 128 columns, 4 waves (each 64 x 32).  A tile that spans two groups handles them one after the other, rows of the other
// group zeroed in the A image.
struct OutlierSideArgs {
    const f16_t* x16; long ldx;              // unquantised activations [M][K]
    const int8_t* cb; const float* scb;      // weights [N][K] row-wise int8 + row absmax
    const int8_t* cbt;                       // ... or (cb == NULL) the fragment-tiled copy (i8_tiled_row_off / i8_tiled_k_off)
    const f16_t* tmp; const f16_t* R; long ldr; f16_t* C; long ldc;   // v in, residual in, out (tmp has C's pitch)
    const int* oc_cnt; const int* oc_list; int oc_ld, thr;
    const int* row_group; int group_div; int row_off;     // group of row m = row_group[(m + row_off) / group_div], as the GEMM's epilogue (int8_util.h i8_row)
    int M, N, K;
};
__global__ __launch_bounds__(256) void i8_outlier_side_kernel(OutlierSideArgs a) {
    constexpr int XP = 80, RM = 64, RN = 128;                       // LDS row pitch in bytes (64 B of k + 16 B skew), tile rows / columns
    __shared__ __attribute__((aligned(16))) char sx[RM * XP];
    __shared__ __attribute__((aligned(16))) char sw[RN * XP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * RM, n0 = blockIdx.y * RN;
    const int mlast = min(m0 + RM, a.M) - 1;
    auto group_of = [&](int m) { return a.row_group ? a.row_group[(m + a.row_off) / a.group_div] : (m + a.row_off) / a.group_div; };
    const int g_first = group_of(m0), g_last = group_of(mlast);     // group ids do not decrease with the row
    for (int g = g_first; g <= g_last; ++g) {
        const int cnt = a.oc_cnt[g];
        if (cnt <= a.thr) continue;                                  // (block-uniform: the epilogue of the GEMM finished these rows)
        const int* lst = a.oc_list + (long)g * a.oc_ld;
        f32x4 acc[4][2];                                             // [m block][n block] of this wave's 64 x 32
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < cnt; k0 += 32) {
            __syncthreads();
            // gather: X chunk [64 rows][32 listed columns], W chunk [128 rows][32] dequantised to fp16 values
            for (int i = tid; i < RM * 32; i += 256) {
                const int r = i >> 5, c = i & 31, m = m0 + r;
                f16_t v = (f16_t)0.f;
                if (k0 + c < cnt && m < a.M && group_of(m) == g) v = a.x16[(long)m * a.ldx + lst[k0 + c]];
                *(f16_t*)(sx + r * XP + c * 2) = v;
            }
            for (int i = tid; i < RN * 32; i += 256) {
                const int r = i >> 5, c = i & 31, n = n0 + r;
                f16_t v = (f16_t)0.f;
                if (k0 + c < cnt && n < a.N) v = (f16_t)rT<f16_t>(__fmul_rn(__fmul_rn((float)(a.cb ? a.cb[(long)n * a.K + lst[k0 + c]] : a.cbt[i8_tiled_row_off(n, a.K) + i8_tiled_k_off(lst[k0 + c])]), a.scb[n]), INT8_DEQ_W));
                *(f16_t*)(sw + r * XP + c * 2) = v;
            }
            __syncthreads();
            f16x8 xf[4], wf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) xf[i] = *(const f16x8*)(sx + (i * 16 + fr) * XP + fg * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *(const f16x8*)(sw + (wid * 32 + j * 16 + fr) * XP + fg * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        // D[n = n0 + wid*32 + j*16 + fg*4 + e][m = m0 + i*16 + fr]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i * 16 + fr;
            if (m >= a.M || group_of(m) != g) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wid * 32 + j * 16 + fg * 4;
                if (n >= a.N) continue;
                const f16x4 v = *(const f16x4*)(a.tmp + (long)m * a.ldc + n);
                f16x4 o;
                if (a.R) {
                    const f16x4 rv = *(const f16x4*)(a.R + (long)m * a.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16_t)(rT<f16_t>(__fadd_rn((float)v[e], acc[i][j][e])) + (float)rv[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16_t)__fadd_rn((float)v[e], acc[i][j][e]);
                }
                *(f16x4*)(a.C + (long)m * a.ldc + n) = o;
            }
        }
    }
}
void launch_i8_outlier_side(const GemmArgs& g, hipStream_t s) {
    if (!g.q.defer_out || g.M <= 0) return;
    OutlierSideArgs a{};
    a.x16 = (const f16_t*)g.q.x16; a.ldx = g.q.ldx16; a.cb = g.w_tiled ? nullptr : (const int8_t*)g.W; a.cbt = g.w_tiled ? (const int8_t*)g.W : nullptr; a.scb = g.q.scb;
    a.tmp = (const f16_t*)g.q.defer_out; a.R = (const f16_t*)g.R; a.ldr = g.ldr; a.C = (f16_t*)g.C; a.ldc = g.ldc;
    a.oc_cnt = g.q.oc_cnt; a.oc_list = g.q.oc_list; a.oc_ld = g.q.oc_ld; a.thr = g.q.defer_thr;
    a.row_group = g.q.row_group; a.group_div = g.q.group_div; a.row_off = g.q.row_off; a.M = g.M; a.N = g.N; a.K = g.K;
    hipLaunchKernelGGL(i8_outlier_side_kernel, dim3((g.M + 63) / 64, (g.N + 127) / 128), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------- k-major copy of an int8 weight matrix: wt[k][n] = w[n][k]
// (decode step: a consumer that adds a row's outlier columns needs W[:, k] for its output columns; out of the fragment-tiled copy that is one
//  byte per 16-byte chunk - 512 HBM sectors per outlier and row - out of this copy it is 8 consecutive bytes per thread, 32 sectors)
__global__ __launch_bounds__(256) void transpose_i8_kernel(const int8_t* w, int8_t* wt, int N, int K) {
    __shared__ int8_t tile[64][68];
    const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) { const int n = n0 + i, k = k0 + tx; if (n < N && k < K) tile[i][tx] = w[(long)n * K + k]; }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) { const int k = k0 + i, n = n0 + tx; if (k < K && n < N) wt[(long)k * N + n] = tile[tx][i]; }
}
void launch_transpose_i8(const int8_t* w, int8_t* wt, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(transpose_i8_kernel, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, s, w, wt, N, K);
}

// ---------------------------------------------------------------- int8 encoder: V columns of the row-major QKV matrix -> V^T [seg][C][vt_ld]
// (the 16-bit path writes V^T from the QKV GEMM's epilogue; with the dequantisation on top that epilogue spills registers)
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* qkv, long ld, int col0, bf16_t* vt, int T, int C, int vt_ld, long vt_seg_stride) {
    __shared__ bf16_t tile[64][66];
    const int seg = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int t = t0 + i;
        if (t < T && c0 + tx < C) tile[i][tx] = qkv[((long)seg * T + t) * ld + col0 + c0 + tx];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) vt[(long)seg * vt_seg_stride + (long)c * vt_ld + t] = tile[tx][i];
    }
}
void launch_transpose_v(const bf16_t* qkv, long ld, int col0, bf16_t* vt, int n_seg, int T, int C, int vt_ld, long vt_seg_stride, hipStream_t s) {
    hipLaunchKernelGGL(transpose_v_kernel, dim3((T + 63) / 64, (C + 63) / 64, n_seg), dim3(256), 0, s, qkv, ld, col0, vt, T, C, vt_ld, vt_seg_stride);
}
