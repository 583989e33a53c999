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
void launch_quant_act(const QuantActArgs& a, hipStream_t s) {
    if (a.M <= 0) return;
    launch_fill_i32((int*)a.flags, 0, (int)(((long)a.G * a.K + 3) / 4), s);
    hipLaunchKernelGGL(qa_stats_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qa_quant_kernel, dim3((a.M + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(qa_lists_kernel, dim3(a.G), dim3(256), 0, s, a);
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
