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
template <typename KD, typename AccE>
__device__ __forceinline__ float gemm_lin(const GemmArgs& a, AccE accv, int m, int n, float bias, const I8Row& rw, float sb) {
    typedef typename KD::out OT;
    if constexpr (KD::I8) {
        float v = rT<f16_t>(fmaf((float)accv, __fmul_rn(__fmul_rn(rw.sa, sb), MM_DEQUANT_CONST), bias));
        if (rw.cnt > 0) v = i8_add_outliers(a.q, (const int8_t*)a.W + (long)n * a.K, sb, rw.g, rw.cnt, m, v);
        return v;
    } else {
        return rT<OT>((float)accv + bias);
    }
}
