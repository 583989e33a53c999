// LLM.int8 device helpers shared by the decode-step consumers (elementwise.hip, attn.hip) and the int8 GEMM epilogues (gemm_i8.hip).
// Spec: oracle/sonic_oracle.c linear_int8 (bitsandbytes Linear8bitLt, threshold 6.0, asr.py:182-198).
#pragma once
#include "common.h"
#include "kernels.h"

#define LLM_INT8_THRESHOLD 6.0f
#define MM_DEQUANT_CONST 6.200012e-05f          // 1 / (127 * 127), bitsandbytes csrc/kernels.cu
#define INT8_DEQ_W 7.874015718698502e-3f        // 1 / 127, int8_vectorwise_dequant

// int8 weight W[n][k] out of the row-major matrix or, when present, the fragment-tiled copy (element (n, k) at
// ((n/16)*(K/64) + k/64)*1024 + (((k%64)/16)*16 + n%16)*16 + k%16, gemm.hip tile_weights_i8_kernel)
__device__ __forceinline__ float deq_w(const DeqInfo& q, int n, int k) {
    if (q.cbt) return (float)q.cbt[((long)(n >> 4) * (q.K >> 6) + (k >> 6)) * 1024 + ((((k & 63) >> 4) * 16) + (n & 15)) * 16 + (k & 15)];
    return (float)q.cb[(long)n * q.K + k];
}

// four consecutive output columns [col, col+4) of row `row`: int32 slab sum -> dequant -> (+ outlier columns), all as fp16 values.
// Every load (slabs, statistics, outlier count) is issued before the first use: a rolled slab loop costs one L2 round trip per slab.
__device__ __forceinline__ f32x4 deq4(const DeqInfo& q, const float* P, int ks, int mpad, int row, int col, int N) {
    const int* Pi = (const int*)P;
    i32x4 sl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sl[k] = *(const i32x4*)(Pi + ((long)(k < ks ? k : 0) * mpad + row) * N + col);
    const float sa = q.sca[row];
    const f32x4 sb = *(const f32x4*)(q.scb + col);
    const int g = q.row_group ? q.row_group[row / q.group_div] : row / q.group_div;
    const int n = q.oc_cnt[g];
    i32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < ks) acc += sl[k];
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rT<f16_t>(fmaf((float)acc[j], __fmul_rn(__fmul_rn(sa, sb[j]), MM_DEQUANT_CONST), 0.0f));
    if (n > 0) {
        const f16_t* xr = (const f16_t*)q.x16 + (long)row * q.ldx16;
        float a2[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const int k = q.oc_list[(long)g * q.oc_ld + i];
            const float xv = (float)xr[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wdq = rT<f16_t>(__fmul_rn(__fmul_rn(deq_w(q, col + j, k), sb[j]), INT8_DEQ_W));
                a2[j] = __fmaf_rn(xv, wdq, a2[j]);      // the oracle rounds product and sum separately; the product of two fp16 values is exact in fp32, so one FMA gives the same bits
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rT<f16_t>(__fadd_rn(v[j], a2[j]));
    }
    return v;
}

// scalar form of deq4 (one output column): the decode attention prologue owns single columns per thread
__device__ __forceinline__ float deq1(const DeqInfo& q, const float* P, int ks, int mpad, int row, int col, int N) {
    const int* Pi = (const int*)P;
    int sl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sl[k] = Pi[((long)(k < ks ? k : 0) * mpad + row) * N + col];
    const float sa = q.sca[row], sb = q.scb[col];
    const int g = q.row_group ? q.row_group[row / q.group_div] : row / q.group_div;
    const int n = q.oc_cnt[g];
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < ks) acc += sl[k];
    float v = rT<f16_t>(fmaf((float)acc, __fmul_rn(__fmul_rn(sa, sb), MM_DEQUANT_CONST), 0.0f));
    if (n > 0) {
        const f16_t* xr = (const f16_t*)q.x16 + (long)row * q.ldx16;
        float a2 = 0.f;
        for (int i = 0; i < n; ++i) {
            const int k = q.oc_list[(long)g * q.oc_ld + i];
            a2 = __fmaf_rn((float)xr[k], rT<f16_t>(__fmul_rn(__fmul_rn(deq_w(q, col, k), sb), INT8_DEQ_W)), a2);   // (exact product: see deq4)
        }
        v = rT<f16_t>(__fadd_rn(v, a2));
    }
    return v;
}

// ---- decode-step consumers, staged form (round 3) ------------------------------------------------------------------------------
// deq4 / deq1 walk the row's outlier list once per call: count -> list entry -> x value -> weight bytes, three dependent global round
// trips per outlier and per call (the SwiGLU consumer made four calls per thread).  Here a block that owns one row fetches the first
// OUTL_CAP (k, x) pairs ONCE, speculatively and together with the slab loads (one round trip; the producer also stores the x values next
// to the list, QuantOut.oc_val), parks them in LDS, and every thread then walks them with all weight bytes of two outliers in flight.
// Same arithmetic, same order of the per-element sums: bit-identical with deq4 / deq1.
#define OUTL_CAP 64
struct OutlStage { int kk; float xv; int n, g, cap; };   // cap = pairs staged in LDS = min(OUTL_CAP, block size)
// loads only; every thread of the block
__device__ __forceinline__ OutlStage outl_issue(const DeqInfo& q, int row) {
    OutlStage o;
    o.g = q.row_group ? q.row_group[row / q.group_div] : row / q.group_div;
    o.n = q.oc_cnt[o.g];
    o.kk = 0; o.xv = 0.f;
    o.cap = min(OUTL_CAP, (int)blockDim.x);
    const int t = threadIdx.x;
    if (t < OUTL_CAP) {
        const long at = (long)o.g * q.oc_ld + min(t, q.oc_ld - 1);
        o.kk = q.oc_list[at];                              // entries past the count are stale: clamped below, never used
        if (q.oc_val) o.xv = q.oc_val[at];
    }
    return o;
}
// LDS + barrier; every thread of the block.  s_k / s_x: OUTL_CAP entries each.
__device__ __forceinline__ void outl_commit(const DeqInfo& q, int row, const OutlStage& o, int* s_k, float* s_x) {
    const int t = threadIdx.x;
    if (o.n <= 0) return;                              // (block-uniform: the usual case costs neither LDS traffic nor a barrier)
    if (t < OUTL_CAP) {
        const int kk = min(max(o.kk, 0), q.K - 1);
        s_k[t] = kk;
        s_x[t] = q.oc_val ? o.xv : (float)((const f16_t*)q.x16)[(long)row * q.ldx16 + kk];
    }
    __syncthreads();
}
// The same stage when nobody has listed the row's outliers (DeqInfo.scan): the block reads the row of x16 itself - thread t owns the 8-element
// groups [t * GPT, (t + 1) * GPT), so list positions ascend with k exactly as quant_emit_row writes them - and fills the LDS stage; entries
// beyond it spill into the row's oc_list / oc_val space (read back by this block only).  s_i: >= 17 ints of scratch.
template <int GPT> struct OutlScan { f16x8 x[GPT]; };
template <int GPT>
__device__ __forceinline__ OutlScan<GPT> outl_scan_issue(const DeqInfo& q, int row) {
    OutlScan<GPT> r;
    const int ng = q.K >> 3;
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int g8 = min((int)threadIdx.x * GPT + i, ng - 1);
        r.x[i] = *(const f16x8*)((const f16_t*)q.x16 + (long)row * q.ldx16 + g8 * 8);
    }
    return r;
}
template <int GPT>
__device__ __forceinline__ OutlStage outl_scan_commit(const DeqInfo& q, int row, const OutlScan<GPT>& r, int* s_k, float* s_x, int* s_i) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (blockDim.x + 63) >> 6, ng = q.K >> 3;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < GPT; ++i)
        if (tid * GPT + i < ng) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt += !(fabsf((float)r.x[i][j]) < LLM_INT8_THRESHOLD);
        }
    int incl = cnt;
    if (__ballot(cnt > 0)) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    }
    if (lane == 63) s_i[wid] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < nw; ++w) { if (w < wid) base += s_i[w]; total += s_i[w]; }
    OutlStage o;
    o.g = row; o.n = total; o.cap = OUTL_CAP; o.kk = 0; o.xv = 0.f;
    if (total > 0) {                                   // block-uniform
        int pos = base + incl - cnt;
        if (cnt > 0) {
#pragma unroll
            for (int i = 0; i < GPT; ++i)
                if (tid * GPT + i < ng) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float y = (float)r.x[i][j];
                        if (!(fabsf(y) < LLM_INT8_THRESHOLD)) {
                            const int k = (tid * GPT + i) * 8 + j;
                            if (pos < OUTL_CAP) { s_k[pos] = k; s_x[pos] = y; }
                            else { ((int*)q.oc_list)[(long)row * q.oc_ld + pos] = k; ((float*)q.oc_val)[(long)row * q.oc_ld + pos] = y; }
                            ++pos;
                        }
                    }
                }
        }
        __syncthreads();
    }
    return o;
}

// outlier i of the row: from LDS, or (lists longer than OUTL_CAP) from memory
__device__ __forceinline__ void outl_get(const DeqInfo& q, int row, const OutlStage& o, const int* s_k, const float* s_x, int i, int& k, float& xv) {
    if (i < o.cap) { k = s_k[i]; xv = s_x[i]; }
    else {
        k = q.oc_list[(long)o.g * q.oc_ld + i];
        xv = q.scan ? q.oc_val[(long)o.g * q.oc_ld + i] : (float)((const f16_t*)q.x16)[(long)row * q.ldx16 + k];
    }
}
// byte address of W[n][k] (see deq_w); consecutive n inside a 16-row group are `deq_w_stride` bytes apart
__device__ __forceinline__ const int8_t* deq_w_ptr(const DeqInfo& q, int n, int k) {
    if (q.cbk) return q.cbk + (long)k * q.N + n;
    if (q.cbt) return q.cbt + ((long)(n >> 4) * (q.K >> 6) + (k >> 6)) * 1024 + ((((k & 63) >> 4) * 16) + (n & 15)) * 16 + (k & 15);
    return q.cb + (long)n * q.K + k;
}
__device__ __forceinline__ long deq_w_stride(const DeqInfo& q) { return q.cbk ? 1 : q.cbt ? 16 : q.K; }

// NG groups of 8 consecutive output columns (col[g] % 8 == 0) of one row.  KU = slabs requested up front.
template <int NG, int KU> struct Slab8 { i32x4 sl[KU][NG][2]; f32x4 sb[NG][2]; float sa; };
template <int NG, int KU>
__device__ __forceinline__ Slab8<NG, KU> slab8_load(const DeqInfo& q, const float* P, int ks, int mpad, int row, const int (&col)[NG], int N) {
    Slab8<NG, KU> s;
    const int* Pi = (const int*)P;
#pragma unroll
    for (int k = 0; k < KU; ++k)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int* p = Pi + ((long)(k < ks ? k : 0) * mpad + row) * N + col[g];
            s.sl[k][g][0] = *(const i32x4*)p; s.sl[k][g][1] = *(const i32x4*)(p + 4);
        }
    if (q.scan) { const f32x4 pm = *(const f32x4*)(q.sca + row * 4); s.sa = fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3])); }
    else s.sa = q.sca[row];
#pragma unroll
    for (int g = 0; g < NG; ++g) { s.sb[g][0] = *(const f32x4*)(q.scb + col[g]); s.sb[g][1] = *(const f32x4*)(q.scb + col[g] + 4); }
    return s;
}
template <int NG, int KU>
__device__ __forceinline__ void slab8_finish(const DeqInfo& q, const float* P, int ks, int mpad, int row, const int (&col)[NG], int N, const Slab8<NG, KU>& s,
                                             const OutlStage& o, const int* s_k, const float* s_x, float (&v)[NG][8]) {
    const int* Pi = (const int*)P;
    i32x4 acc[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g) { acc[g][0] = (i32x4){0, 0, 0, 0}; acc[g][1] = (i32x4){0, 0, 0, 0}; }
#pragma unroll
    for (int k = 0; k < KU; ++k)
        if (k < ks) {
#pragma unroll
            for (int g = 0; g < NG; ++g) { acc[g][0] += s.sl[k][g][0]; acc[g][1] += s.sl[k][g][1]; }
        }
    for (int k = KU; k < ks; ++k)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int* p = Pi + ((long)k * mpad + row) * N + col[g];
            acc[g][0] += *(const i32x4*)p; acc[g][1] += *(const i32x4*)(p + 4);
        }
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[g][j] = rT<f16_t>(fmaf((float)acc[g][j >> 2][j & 3], __fmul_rn(__fmul_rn(s.sa, s.sb[g][j >> 2][j & 3]), MM_DEQUANT_CONST), 0.0f));
    if (o.n <= 0) return;
    float a2[NG][8];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) a2[g][j] = 0.f;
    const long ws = deq_w_stride(q);
    int i = 0;
    for (; i + 1 < o.n; i += 2) {                      // two outliers' weight bytes in flight
        int k0, k1; float x0, x1;
        outl_get(q, row, o, s_k, s_x, i, k0, x0); outl_get(q, row, o, s_k, s_x, i + 1, k1, x1);
        int8_t w0[NG][8], w1[NG][8];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (q.cbk) {                                 // k-major copy: the 8 columns are 8 consecutive bytes
                const int2 b0 = *(const int2*)(q.cbk + (long)k0 * q.N + col[g]), b1 = *(const int2*)(q.cbk + (long)k1 * q.N + col[g]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    w0[g][j] = (int8_t)(((j < 4 ? b0.x : b0.y) >> ((j & 3) * 8)) & 0xFF);
                    w1[g][j] = (int8_t)(((j < 4 ? b1.x : b1.y) >> ((j & 3) * 8)) & 0xFF);
                }
            } else {
                const int8_t* p0 = deq_w_ptr(q, col[g], k0); const int8_t* p1 = deq_w_ptr(q, col[g], k1);
#pragma unroll
                for (int j = 0; j < 8; ++j) { w0[g][j] = p0[j * ws]; w1[g][j] = p1[j * ws]; }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sb = s.sb[g][j >> 2][j & 3];
                a2[g][j] = __fmaf_rn(x0, rT<f16_t>(__fmul_rn(__fmul_rn((float)w0[g][j], sb), INT8_DEQ_W)), a2[g][j]);
                a2[g][j] = __fmaf_rn(x1, rT<f16_t>(__fmul_rn(__fmul_rn((float)w1[g][j], sb), INT8_DEQ_W)), a2[g][j]);
            }
    }
    if (i < o.n) {
        int k0; float x0;
        outl_get(q, row, o, s_k, s_x, i, k0, x0);
        int8_t w0[NG][8];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (q.cbk) {
                const int2 b0 = *(const int2*)(q.cbk + (long)k0 * q.N + col[g]);
#pragma unroll
                for (int j = 0; j < 8; ++j) w0[g][j] = (int8_t)(((j < 4 ? b0.x : b0.y) >> ((j & 3) * 8)) & 0xFF);
            } else {
                const int8_t* p0 = deq_w_ptr(q, col[g], k0);
#pragma unroll
                for (int j = 0; j < 8; ++j) w0[g][j] = p0[j * ws];
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                a2[g][j] = __fmaf_rn(x0, rT<f16_t>(__fmul_rn(__fmul_rn((float)w0[g][j], s.sb[g][j >> 2][j & 3]), INT8_DEQ_W)), a2[g][j]);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[g][j] = rT<f16_t>(__fadd_rn(v[g][j], a2[g][j]));
}

// two single columns (col, col + dcol) of one row: the decode attention prologue
struct Slab1x2 { int sl[8][2]; float sb[2]; float sa; };
__device__ __forceinline__ Slab1x2 slab1x2_load(const DeqInfo& q, const float* P, int ks, int mpad, int row, int col, int dcol, int N) {
    Slab1x2 s;
    const int* Pi = (const int*)P;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int* p = Pi + ((long)(k < ks ? k : 0) * mpad + row) * N + col;
        s.sl[k][0] = p[0]; s.sl[k][1] = p[dcol];
    }
    s.sa = q.sca[row]; s.sb[0] = q.scb[col]; s.sb[1] = q.scb[col + dcol];
    return s;
}
__device__ __forceinline__ void slab1x2_finish(const DeqInfo& q, const float* P, int ks, int mpad, int row, int col, int dcol, int N, const Slab1x2& s,
                                               const OutlStage& o, const int* s_k, const float* s_x, float& v0, float& v1) {
    const int* Pi = (const int*)P;
    int acc[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < ks) { acc[0] += s.sl[k][0]; acc[1] += s.sl[k][1]; }
    for (int k = 8; k < ks; ++k) { const int* p = Pi + ((long)k * mpad + row) * N + col; acc[0] += p[0]; acc[1] += p[dcol]; }
    float v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) v[c] = rT<f16_t>(fmaf((float)acc[c], __fmul_rn(__fmul_rn(s.sa, s.sb[c]), MM_DEQUANT_CONST), 0.0f));
    if (o.n > 0) {
        float a2[2] = {0.f, 0.f};
        int i = 0;
        for (; i + 3 < o.n; i += 4) {                  // four outliers' weight bytes in flight
            int k[4]; float x[4]; int8_t w[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) outl_get(q, row, o, s_k, s_x, i + u, k[u], x[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) { w[u][0] = *deq_w_ptr(q, col, k[u]); w[u][1] = *deq_w_ptr(q, col + dcol, k[u]); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) a2[c] = __fmaf_rn(x[u], rT<f16_t>(__fmul_rn(__fmul_rn((float)w[u][c], s.sb[c]), INT8_DEQ_W)), a2[c]);
        }
        for (; i < o.n; ++i) {
            int k; float x;
            outl_get(q, row, o, s_k, s_x, i, k, x);
            const int8_t w0 = *deq_w_ptr(q, col, k), w1 = *deq_w_ptr(q, col + dcol, k);
            a2[0] = __fmaf_rn(x, rT<f16_t>(__fmul_rn(__fmul_rn((float)w0, s.sb[0]), INT8_DEQ_W)), a2[0]);
            a2[1] = __fmaf_rn(x, rT<f16_t>(__fmul_rn(__fmul_rn((float)w1, s.sb[1]), INT8_DEQ_W)), a2[1]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) v[c] = rT<f16_t>(__fadd_rn(v[c], a2[c]));
    }
    v0 = v[0]; v1 = v[1];
}

// the rare part of an int8 GEMM epilogue (inlined: an out-of-line call made every instantiation spill around the call site):
// v + sum over the outlier columns of the row's group of x[m][k] * fp16(CB[n][k] * SCB[n] / 127), rounded to fp16
__device__ __forceinline__ float i8_add_outliers(const GemmI8& q, const int8_t* wr, float sb, int g, int cnt, long m, float v) {
    const f16_t* xr = (const f16_t*)q.x16 + m * q.ldx16;
    float a2 = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const int k = q.oc_list[(long)g * q.oc_ld + i];
        a2 = __fmaf_rn((float)xr[k], rT<f16_t>(__fmul_rn(__fmul_rn((float)wr[k], sb), INT8_DEQ_W)), a2);         // (exact product: see deq4)
    }
    return rT<f16_t>(__fadd_rn(v, a2));
}

// W[n][k] from the fragment-tiled copy: the prefill GEMMs of the decoder read the decode step's tiled weights since round 5 (their row-major int8 copy is gone) and,
// since round 6, gather the outlier columns from it too (rounds 3 - 5 kept a k-major copy for that: 1.29 GB at full size).  wt = W + i8_tiled_row_off(n, K), element k at
// i8_tiled_k_off(k).  Compiled only into the epilogues that can meet such an operand (gemm_lin<KD, WK>).
__device__ __forceinline__ float i8_add_outliers_t(const GemmI8& q, const int8_t* wt, float sb, int g, int cnt, long m, float v) {
    const f16_t* xr = (const f16_t*)q.x16 + m * q.ldx16;
    float a2 = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const int k = q.oc_list[(long)g * q.oc_ld + i];
        a2 = __fmaf_rn((float)xr[k], rT<f16_t>(__fmul_rn(__fmul_rn((float)wt[i8_tiled_k_off(k)], sb), INT8_DEQ_W)), a2);
    }
    return rT<f16_t>(__fadd_rn(v, a2));
}

// Per-row metadata of an int8 GEMM epilogue, loaded once per row a lane touches (not per element)
struct I8Row { float sa; int g, cnt; bool defer; };
template <typename KD>
__device__ __forceinline__ I8Row i8_row(const GemmArgs& a, int m) {
    I8Row r{0.f, 0, 0, false};
    if constexpr (KD::I8) {
        r.sa = a.q.sca[m];
        r.g = a.q.row_group ? a.q.row_group[(m + a.q.row_off) / a.q.group_div] : (m + a.q.row_off) / a.q.group_div;
        r.cnt = a.q.oc_cnt[r.g];
        if (a.q.defer_out && r.cnt > a.q.defer_thr) { r.defer = true; r.cnt = 0; }     // the side kernel adds this row's outlier columns
    }
    return r;
}
// Linear output of one accumulator element as a float that is exactly representable in the output type: 16-bit kinds round
// acc + bias once; the int8 kind applies the LLM.int8 dequantisation (sonic_oracle.c linear_int8) and adds the outlier columns.
// rw: the row's metadata (i8_row), sb: SCB[n] (int8 kind only).
// WK: the epilogue may meet a fragment-tiled W (GemmArgs.w_tiled: decoder projections; never the GELU epilogue - compiled into it, the second addressing form made the
// encoder's int8 fc1 kernel keep its accumulators in scratch).
template <typename KD, bool WK = true, typename AccE>
__device__ __forceinline__ float gemm_lin(const GemmArgs& a, AccE accv, int m, int n, float bias, const I8Row& rw, float sb) {
    typedef typename KD::out OT;
    if constexpr (KD::I8) {
        float v = rT<f16_t>(fmaf((float)accv, __fmul_rn(__fmul_rn(rw.sa, sb), MM_DEQUANT_CONST), bias));
        if (rw.cnt > 0) {
            // (two addressing forms only: a third one - the k-major copy of rounds 3 - 5 beside these - put 528 bytes of the 256x256 int8 kernels' registers into scratch)
            if (WK && a.w_tiled) v = i8_add_outliers_t(a.q, (const int8_t*)a.W + i8_tiled_row_off(n, a.K), sb, rw.g, rw.cnt, m, v);
            else v = i8_add_outliers(a.q, (const int8_t*)a.W + (long)n * a.K, sb, rw.g, rw.cnt, m, v);
        }
        return v;
    } else {
        return rT<OT>((float)accv + bias);
    }
}
