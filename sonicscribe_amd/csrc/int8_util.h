// LLM.int8 device helpers shared by the decode-step consumers (elementwise.hip, attn.hip) and the int8 GEMM epilogues (gemm_i8.hip).
// Spec: oracle/sonic_oracle.c linear_int8 (bitsandbytes Linear8bitLt, threshold 6.0, asr.py:182-198).
#pragma once
#include "common.h"
#include "kernels.h"

#define LLM_INT8_THRESHOLD 6.0f
#define MM_DEQUANT_CONST 6.200012e-05f          // 1 / (127 * 127), bitsandbytes csrc/kernels.cu
#define INT8_DEQ_W 7.874015718698502e-3f        // 1 / 127, int8_vectorwise_dequant

// four consecutive output columns [col, col+4) of row `row`: int32 slab sum -> dequant -> (+ outlier columns), all as fp16 values
__device__ __forceinline__ f32x4 deq4(const DeqInfo& q, const float* P, int ks, int mpad, int row, int col, int N) {
    const int* Pi = (const int*)P;
    i32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < ks; ++k) acc += *(const i32x4*)(Pi + ((long)k * mpad + row) * N + col);
    const float sa = q.sca[row];
    const f32x4 sb = *(const f32x4*)(q.scb + col);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rT<f16_t>(fmaf((float)acc[j], __fmul_rn(__fmul_rn(sa, sb[j]), MM_DEQUANT_CONST), 0.0f));
    const int g = q.row_group ? q.row_group[row / q.group_div] : row / q.group_div;
    const int n = q.oc_cnt[g];
    if (n > 0) {
        const f16_t* xr = (const f16_t*)q.x16 + (long)row * q.ldx16;
        float a2[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const int k = q.oc_list[(long)g * q.oc_ld + i];
            const float xv = (float)xr[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wdq = rT<f16_t>(__fmul_rn(__fmul_rn((float)q.cb[(long)(col + j) * q.K + k], sb[j]), INT8_DEQ_W));
                a2[j] = __fadd_rn(a2[j], __fmul_rn(xv, wdq));      // (no FMA contraction: the oracle rounds product and sum separately)
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rT<f16_t>(__fadd_rn(v[j], a2[j]));
    }
    return v;
}


// Linear output of one accumulator element as a float that is exactly representable in the output type: 16-bit kinds round
// acc + bias once; the int8 kind applies the LLM.int8 dequantisation (sonic_oracle.c linear_int8) and adds the outlier columns.
// the rare part of an int8 GEMM epilogue (inlined: an out-of-line call made every instantiation spill around the call site):
// v + sum over the outlier columns of the row's group of x[m][k] * fp16(CB[n][k] * SCB[n] / 127), rounded to fp16
__device__ __forceinline__ float i8_add_outliers(const GemmI8& q, const int8_t* wr, float sb, int g, int cnt, long m, float v) {
    const f16_t* xr = (const f16_t*)q.x16 + m * q.ldx16;
    float a2 = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const int k = q.oc_list[(long)g * q.oc_ld + i];
        a2 = __fadd_rn(a2, __fmul_rn((float)xr[k], rT<f16_t>(__fmul_rn(__fmul_rn((float)wr[k], sb), INT8_DEQ_W))));
    }
    return rT<f16_t>(__fadd_rn(v, a2));
}

template <typename KD, typename AccE>
__device__ __forceinline__ float gemm_lin(const GemmArgs& a, AccE accv, int m, int n, float bias) {
    typedef typename KD::out OT;
    if constexpr (KD::I8) {
        const float sb = a.q.scb[n];
        float v = rT<f16_t>(fmaf((float)accv, __fmul_rn(__fmul_rn(a.q.sca[m], sb), MM_DEQUANT_CONST), bias));
        const int g = a.q.row_group ? a.q.row_group[m / a.q.group_div] : m / a.q.group_div;
        const int cnt = a.q.oc_cnt[g];
        if (cnt > 0) v = i8_add_outliers(a.q, (const int8_t*)a.W + (long)n * a.K, sb, g, cnt, m, v);
        return v;
    } else {
        return rT<OT>((float)accv + bias);
    }
}
