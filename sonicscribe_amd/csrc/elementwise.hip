// Memory-bound glue kernels: LayerNorm / RMSNorm with wavefront
// reductions, rotate-half RoPE (partial-32 encoder, full-128 decoder) fused with the KV-cache append,
// the decode-step consumers of the skinny-GEMM partial slabs, embedding gather, fused argmax + greedy
// controller, synthetic weight generator.  All 16-bit traffic is 16 B per lane (8 elements).
//
// Rounding boundaries reproduce torch's op sequence of the reference path: every torch op output is rounded to the
// activation dtype once, arithmetic inside an op is fp32.  Kernels are templated on that dtype T: bf16 (`mode="native"`) or
// IEEE half (`mode="int8"`, asr.py:61,296).  In int8 mode the decode-step consumers read int32 slabs of a quantised skinny
// GEMM (deq4) and producers that own whole rows also emit them quantised for the next Linear8bitLt (quant_emit_row).
#include "common.h"
#include "kernels.h"

#include "int8_util.h"

// A block that owns one whole row (thread c holds its elements [8c, 8c+8) as fp16 values in y, threads with !active hold nothing)
// emits the row quantised: absmax without the elements >= 6.0, int8 = rn(y * 127 / absmax) (0 for outliers), and the ascending
// list of the outlier positions.  Every thread of the block must call it.  s_f: >= 16 floats, s_i: >= 17 ints of LDS scratch.
// FRESH: nobody has touched s_f / s_i in this kernel before (no barrier needed in front of the first write).
template <bool FRESH = false>
__device__ __forceinline__ void quant_emit_row(const float (&y)[8], bool active, int c, int row, const QuantOut& qo, float* s_f, int* s_i) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (blockDim.x + 63) >> 6;
    float amax = -1.17549435e-38f;
    int cnt = 0;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float a = fabsf(y[j]); if (a < LLM_INT8_THRESHOLD) amax = fmaxf(amax, a); else ++cnt; }
    }
    amax = wave_max(amax);
    int incl = cnt;                                   // inclusive scan of the outlier counts inside the wave (only where the wave holds any)
    if (__ballot(cnt > 0)) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    }
    if constexpr (!FRESH) __syncthreads();           // scratch may still be in use by the caller
    if (lane == 63) s_i[wid] = incl;
    if (lane == 0) s_f[wid] = amax;
    __syncthreads();
    float bm = s_f[0]; int base = 0, total = 0;
    for (int w = 0; w < nw; ++w) { bm = fmaxf(bm, s_f[w]); if (w < wid) base += s_i[w]; total += s_i[w]; }
    const float scale = 127.0f / bm;
    if (tid == 0) { qo.sca[row] = bm; qo.oc_cnt[row] = total; }
    if (active) {
        int pos = base + incl - cnt;
        int pk[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool out = !(fabsf(y[j]) < LLM_INT8_THRESHOLD);
            int qv = (out || !(bm > 0.f)) ? 0 : (int)rintf(y[j] * scale);
            pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
            if (out) {
                const long at = (long)row * qo.oc_ld + pos++;
                qo.oc_list[at] = c * 8 + j;
                if (qo.oc_val) qo.oc_val[at] = y[j];
            }
        }
        *(int2*)(qo.q + (long)row * qo.ldq + c * 8) = make_int2(pk[0], pk[1]);
    }
}

// ---------------------------------------------------------------- LayerNorm (modeling_glmasr.py:246-247,305)
// one wave per row, d % 8 == 0, d <= 2048; two-pass in registers (mean, then centred variance).
// qa.q != null (int8 mode, the row feeds a Linear8bitLt): the wave that normalised the row also emits what the first two passes of
// launch_quant_act would compute from it - row absmax without the elements >= 6.0, int8 codes (those elements as 0), and their group flags.
template <typename T, bool Q>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, const float* w, const float* b, T* y, int rows, int d, float eps, QuantActArgs qa) {
    typedef typename ET<T>::v8 V8;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (long)row * d;
    const int nv = d >> 3;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const V8 t = *(const V8*)(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)t[j]; s += v[i][j]; }
        }
    }
    const float mean = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + i * 64 < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float c = v[i][j] - mean; q += c * c; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / d + eps);
    float amax = -1.17549435e-38f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (T)(((v[i][j] - mean) * rstd) * w[c * 8 + j] + b[c * 8 + j]);
                if constexpr (Q) {
                    v[i][j] = (float)o[j];
                    const float av = fabsf(v[i][j]);
                    if (av < LLM_INT8_THRESHOLD) amax = fmaxf(amax, av);
                }
            }
            *(V8*)(y + (long)row * d + c * 8) = o;
        }
    }
    if constexpr (!Q) return;
    amax = wave_max(amax);
    if (lane == 0) qa.sca[row] = amax;
    const int g = qa.gmap ? qa.gmap[row / qa.gdiv] : row / qa.gdiv;
    unsigned char* fl = qa.flags + (long)g * qa.K;
    const float scale = 127.0f / amax;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            int pk[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool out = !(fabsf(v[i][j]) < LLM_INT8_THRESHOLD);
                if (out) fl[c * 8 + j] = 1;                       // (benign race: every writer stores the same value)
                const int qv = (out || !(amax > 0.f)) ? 0 : (int)rintf(v[i][j] * scale);
                pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
            }
            *(int2*)(qa.q + (long)row * qa.K + c * 8) = make_int2(pk[0], pk[1]);
        }
    }
}

// ---------------------------------------------------------------- RMSNorm (modeling_llama.py:60-65)
template <typename T, bool Q>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* x, const float* w, T* y, int rows, int d, float eps,
                                                      const int* row_map /* optional gather: y[r] = norm(x[row_map[r]]) */, QuantActArgs qa) {
    typedef typename ET<T>::v8 V8;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (long)(row_map ? row_map[row] : row) * d;
    const int nv = d >> 3;
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const V8 t = *(const V8*)(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)t[j]; s += v[i][j] * v[i][j]; }
        }
    }
    const float r = 1.0f / sqrtf(wave_sum(s) / d + eps);
    T* yr = y + (long)row * d;
    float amax = -1.17549435e-38f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (T)(w[c * 8 + j] * rT<T>(v[i][j] * r));
                if constexpr (Q) {
                    v[i][j] = (float)o[j];
                    const float av = fabsf(v[i][j]);
                    if (av < LLM_INT8_THRESHOLD) amax = fmaxf(amax, av);
                }
            }
            *(V8*)(yr + c * 8) = o;
        }
    }
    if constexpr (!Q) return;
    // int8 mode, the row feeds a Linear8bitLt: absmax, codes and group flags in the same pass (as layernorm_kernel)
    amax = wave_max(amax);
    if (lane == 0) qa.sca[row] = amax;
    const int g = qa.gmap ? qa.gmap[row / qa.gdiv] : row / qa.gdiv;
    unsigned char* fl = qa.flags + (long)g * qa.K;
    const float scale = 127.0f / amax;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            int pk[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool out = !(fabsf(v[i][j]) < LLM_INT8_THRESHOLD);
                if (out) fl[c * 8 + j] = 1;
                const int qv = (out || !(amax > 0.f)) ? 0 : (int)rintf(v[i][j] * scale);
                pk[j >> 2] |= (qv & 0xFF) << ((j & 3) * 8);
            }
            *(int2*)(qa.q + (long)row * qa.K + c * 8) = make_int2(pk[0], pk[1]);
        }
    }
}

// decode: x[r] = T(x[r] + T(sum_ks P[ks][r][:])); y[r] = rmsnorm(x[r]) (o_proj / down_proj consumer).
// one block of d/8 threads per row (the step has only <= 64 rows: parallelism comes from the row width).
// int8 mode (dq.sca != null): P holds the int32 slabs of the quantised projection; the normalised row is also emitted quantised (qo.q).
template <typename T>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(T* x, const float* P, int ksplit, int mpad, const float* w, T* y,
                                                          int rows, int d, float eps, DeqInfo dq, QuantOut qo, PrefetchRange pf) {
    typedef typename ET<T>::v8 V8;
    if ((int)blockIdx.x >= rows) {                    // idle-CU prefetch blocks (experiment, option decode_prefetch bit 2): stream the next q|k|v's weights and leave
        prefetch_share(pf, blockIdx.x - rows, gridDim.x - rows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    __shared__ float part[16], qpart[16];             // (qpart / qparti: the re-quantisation's own scratch - no barrier before its first write)
    __shared__ int parti[17], qparti[17];
    __shared__ int s_ok[OUTL_CAP];
    __shared__ float s_ox[OUTL_CAP];
    const int row = blockIdx.x, c = threadIdx.x, lane = c & 63, wid = c >> 6;
    const int nv = d >> 3;
    float v[8];
    float s = 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    T* xr = x + (long)row * d + (c < nv ? c : 0) * 8;
    // the norm weight is requested FIRST: it is cold every step (8 KiB per layer, read once per token) and used only behind the block's
    // reduction - loaded where it is used, it was a second HBM round trip of the kernel's critical path
    f32x4 nw0 = *(const f32x4*)(w + (c < nv ? c : 0) * 8), nw1 = *(const f32x4*)(w + (c < nv ? c : 0) * 8 + 4);
    const V8 t = *(const V8*)xr;                      // (requested before the slabs: one round trip for both)
    if (dq.sca) {
        // int8: the row's outlier pairs and this thread's slabs are requested together, the pairs go through LDS (int8_util.h)
        const int col[1] = {c < nv ? c * 8 : 0};
        float a[1][8];
        if (dq.dbg & 1) {
            const Slab8<1, 8> sl = slab8_load<1, 8>(dq, P, ksplit, mpad, row, col, d);
            OutlStage os{}; os.n = 0;
            slab8_finish<1, 8>(dq, P, ksplit, mpad, row, col, d, sl, os, s_ok, s_ox, a);
        } else if (dq.scan) {
            // the projection quantised its input on the fly: this block lists the input row's outliers itself (K <= 4 * 8 * blockDim.x)
            // (the producers' per-block counts say whether there is anything to find: usually not, and then the block scan and its barriers
            //  are skipped; the row itself is requested either way, before the verdict is known)
            i32x4 nb4 = {1, 0, 0, 0};
            if (dq.scan_cnt) nb4 = *(const i32x4*)(dq.scan_cnt + row * 4);
            const OutlScan<4> sc = outl_scan_issue<4>(dq, row);
            const Slab8<1, 8> sl = slab8_load<1, 8>(dq, P, ksplit, mpad, row, col, d);
            OutlStage os{}; os.n = 0;
            if ((nb4[0] | nb4[1] | nb4[2] | nb4[3]) != 0) os = outl_scan_commit<4>(dq, row, sc, s_ok, s_ox, parti);     // (block-uniform)
            slab8_finish<1, 8>(dq, P, ksplit, mpad, row, col, d, sl, os, s_ok, s_ox, a);
        } else {
            const OutlStage os = outl_issue(dq, row);
            const Slab8<1, 8> sl = slab8_load<1, 8>(dq, P, ksplit, mpad, row, col, d);
            outl_commit(dq, row, os, s_ok, s_ox);
            slab8_finish<1, 8>(dq, P, ksplit, mpad, row, col, d, sl, os, s_ok, s_ox, a);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = a[0][j];
    }
    if (c < nv) {
        if (!dq.sca) {
            // every slab load is issued before the first add (a rolled ksplit loop costs one L2 round trip per slab); the sum keeps its
            // fixed order ks = 0, 1, ...
            f32x4 a0[8], a1[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const float* p = P + ((long)(ks < ksplit ? ks : 0) * mpad + row) * d + c * 8;
                a0[ks] = *(const f32x4*)p; a1[ks] = *(const f32x4*)(p + 4);
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                if (ks < ksplit) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[j] += a0[ks][j]; acc[4 + j] += a1[ks][j]; }
                }
            for (int ks = 8; ks < ksplit; ++ks) {                         // (not reached by the current tilings: ksplit <= 8)
                const float* p = P + ((long)ks * mpad + row) * d + c * 8;
                const f32x4 b0 = *(const f32x4*)p, b1 = *(const f32x4*)(p + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] += b0[j]; acc[4 + j] += b1[j]; }
            }
        }
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = (T)((float)t[j] + rT<T>(acc[j])); v[j] = (float)o[j]; s += v[j] * v[j]; }
        *(V8*)xr = o;
    }
    asm volatile("" : "+v"(nw0), "+v"(nw1));          // (keeps the weight loads above the barrier)
    s = wave_sum(s);
    if (lane == 0) part[wid] = s;
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += part[i];
    const float r = 1.0f / sqrtf(tot / d + eps);
    float yo[8];
    if (c < nv) {
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = (T)((j < 4 ? nw0[j & 3] : nw1[j & 3]) * rT<T>(v[j] * r)); yo[j] = (float)o[j]; }
        *(V8*)(y + (long)row * d + c * 8) = o;
    }
    if (qo.q && !(dq.dbg & 2)) quant_emit_row<true>(yo, c < nv, c, row, qo, qpart, qparti);
}

// decode, 33 .. 64 rows: RMSNorm of the residual rows from the fused o_proj kernel's sum-of-squares partials - the arithmetic skinny_gu_kernel<NORM> does while it
// stages X (same association of the partials: lane half h adds groups 16 h .. 16 h + 15 of a 2048-wide row in ascending order, then the halves are
// added; same two roundings per element), once per row instead of once per gate/up block: at 64 rows that normalisation was ~9 us of VALU time on every
// SIMD of 256 CUs.  One block per row, D / 8 threads.  SS: [region = row / 32][D / 64 groups][32 rows][4].
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_ss_kernel(const T* x, const float* SS, const float* w, T* y, int rows, int d, float eps) {
    typedef typename ET<T>::v8 V8;
    __shared__ float s_half[2], s_scale;
    const int row = blockIdx.x, c = threadIdx.x, nv = d >> 3, ng = d >> 6, ssn = ng >> 1;
    V8 t;
    f32x4 w0, w1;
    if (c < nv) { t = *(const V8*)(x + (long)row * d + c * 8); w0 = *(const f32x4*)(w + c * 8); w1 = *(const f32x4*)(w + c * 8 + 4); }
    if (c < 2) {
        const float* sp = SS + ((long)(row >> 5) * ng * 32 + (long)c * ssn * 32 + (row & 31)) * 4;
        float th = 0.f;
        for (int i = 0; i < ssn; ++i) { const f32x4 v = *(const f32x4*)(sp + (long)i * 128); th += v[0]; th += v[1]; th += v[2]; th += v[3]; }
        s_half[c] = th;
    }
    __syncthreads();
    if (c == 0) { const float tt = s_half[0] + s_half[1]; s_scale = 1.0f / sqrtf(tt / (float)d + eps); }
    __syncthreads();
    if (c < nv) {
        const float rr = s_scale;
        V8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = (T)(w0[j] * rT<T>((float)t[j] * rr)); o[4 + j] = (T)(w1[j] * rT<T>((float)t[4 + j] * rr)); }
        *(V8*)(y + (long)row * d + c * 8) = o;
    }
}
void launch_rmsnorm_ss(const bf16_t* x, const float* SS, const float* w, bf16_t* y, int rows, int d, float eps, hipStream_t s, int dt) {
    DT_SWITCH(dt, T, hipLaunchKernelGGL(rmsnorm_ss_kernel<T>, dim3(rows), dim3(((d >> 3) + 63) / 64 * 64), 0, s, (const T*)x, SS, w, (T*)y, rows, d, eps));
}

// decode: act[r][c] = T(T(silu(T g)) * T u), gate/up rows interleaved in 16-row groups (as EPI_SWIGLU) or, gu8, in 8-row groups (the fused
// gate/up kernel's weight copy, launch_tile_weights_gu8: the one decode copy of the projection since round 5)
template <typename T>
__global__ void swiglu_slab_kernel(const float* P, int ksplit, int mpad, int n2 /* 2*ff */, T* act, int rows, int gu8) {
    typedef typename ET<T>::v4 V4;
    const int ff = n2 >> 1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 outputs
    if (idx >= (long)rows * (ff >> 2)) return;
    const int row = idx / (ff >> 2), c4 = (idx % (ff >> 2)) * 4;
    const int ng = gu8 ? (c4 >> 3) * 16 + (c4 & 7) : (c4 >> 4) * 32 + (c4 & 15), nu = ng + (gu8 ? 8 : 16);
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < ksplit; ++ks) {
        const float* p = P + ((long)ks * mpad + row) * n2;
        g += *(const f32x4*)(p + ng);
        u += *(const f32x4*)(p + nu);
    }
    V4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (T)(rT<T>(silu_f(rT<T>(g[j]))) * rT<T>(u[j]));
    *(V4*)(act + (long)row * ff + c4) = o;
}

// int8 decode: one block (1024 threads) per row; slabs hold the int32 products of the quantised gate and up projections, weight rows
// interleaved in 16-row groups like the prefill GEMM's SwiGLU layout (one int8 copy serves both).  act = fp16(fp16(silu(g)) * u) and,
// in the same pass, the row quantised for down_proj's Linear8bitLt.  ff <= 8192.
__global__ __launch_bounds__(1024) void swiglu_quant_kernel(const float* P, int ksplit, int mpad, int ff, f16_t* act, DeqInfo dq, QuantOut qo) {
    __shared__ float part[16];
    __shared__ int parti[17];
    __shared__ int s_ok[OUTL_CAP];
    __shared__ float s_ox[OUTL_CAP];
    const int row = blockIdx.x, c = threadIdx.x;
    const bool active = c * 8 < ff;
    // act columns [8c, 8c + 8) = gate columns [base, base + 8) and up columns [base + 16, base + 24) of the interleaved projection
    const int base = active ? (c >> 1) * 32 + (c & 1) * 8 : 0;
    const int col[2] = {base, base + 16};
    const OutlStage os = outl_issue(dq, row);
    const Slab8<2, 2> sl = slab8_load<2, 2>(dq, P, ksplit, mpad, row, col, 2 * ff);
    outl_commit(dq, row, os, s_ok, s_ox);
    float gu[2][8];
    slab8_finish<2, 2>(dq, P, ksplit, mpad, row, col, 2 * ff, sl, os, s_ok, s_ox, gu);
    float y[8];
    if (active) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = (f16_t)(rT<f16_t>(silu_f(gu[0][j])) * gu[1][j]); y[j] = (float)o[j]; }
        *(f16x8*)(act + (long)row * ff + c * 8) = o;
    }
    quant_emit_row<true>(y, active, c, row, qo, part, parti);
}

// decode flavour of the activation quantiser: one block per row of an fp16 matrix (K <= 8192)
__global__ __launch_bounds__(1024) void quant_rows_kernel(const f16_t* X, long ld, int K, QuantOut qo) {
    __shared__ float part[16];
    __shared__ int parti[17];
    const int row = blockIdx.x, c = threadIdx.x;
    const bool active = c * 8 < K;
    float y[8];
    if (active) {
        const f16x8 t = *(const f16x8*)(X + (long)row * ld + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (float)t[j];
    }
    quant_emit_row<true>(y, active, c, row, qo, part, parti);
}

// ---------------------------------------------------------------- encoder RoPE (modeling_glmasr.py:153-168)
// in place on the fused q|k buffer [M][ld]; first `rd` dims of each 64-dim head, pairs (i, i + rd/2);
// cs table [T][rd] = cos[0..rd/2) | sin[0..rd/2) (T-rounded fp32).  one thread = 8 pairs.
template <typename T>
__global__ void rope_enc_kernel(T* qk, long ld, int M, int Tn, int heads2 /* q heads + k heads */, int hd, int rd, const float* cs) {
    typedef typename ET<T>::v8 V8;
    const int half = rd >> 1, per_head = half >> 3;  // threads per head
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * heads2 * per_head) return;
    const int u = idx % per_head, hh = (idx / per_head) % heads2;
    const int m = idx / ((long)per_head * heads2), t = m % Tn;
    T* p = qk + (long)m * ld + hh * hd + u * 8;
    const V8 a = *(const V8*)p, bb = *(const V8*)(p + half);
    const float* c = cs + (long)t * rd + u * 8;
    const float* s = c + half;
    V8 o1, o2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x1 = (float)a[j], x2 = (float)bb[j];
        o1[j] = (T)(rT<T>(x1 * c[j]) + rT<T>(-x2 * s[j]));
        o2[j] = (T)(rT<T>(x2 * c[j]) + rT<T>(x1 * s[j]));
    }
    *(V8*)p = o1;
    *(V8*)(p + half) = o2;
}

// ---------------------------------------------------------------- decoder RoPE + KV append (modeling_llama.py:121-143,261-262)
// Source is either the 16-bit QKV matrix of the prefill GEMM or the fp32 slabs of the decode skinny GEMM.
// Writes roped q ([tok][Hq*128]), roped k -> K cache, v -> V cache, and (prefill) v^T -> Vt scratch.
template <typename T, bool SLAB>
__global__ __launch_bounds__(256) void rope_append_kernel(RopeAppendArgs a) {
    typedef typename ET<T>::v8 V8;
    constexpr int HD = 128, HALF = 64;
    const int tok = blockIdx.x;
    const int heads = a.Hq + 2 * a.Hkv;
    const int N = heads * HD;
    const int b = a.tok_seq[tok], pos = a.tok_pos[tok];
    for (int w = threadIdx.x; w < heads * 8; w += blockDim.x) {
        const int hh = w >> 3, u = w & 7;   // head, 8-wide chunk of the first half
        float x1[8], x2[8];
        if (SLAB) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { x1[j] = 0.f; x2[j] = 0.f; }
            for (int ks = 0; ks < a.ksplit; ++ks) {
                const float* p = a.P + ((long)ks * a.mpad + tok) * N + hh * HD + u * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) { x1[j] += p[j]; x2[j] += p[HALF + j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { x1[j] = rT<T>(x1[j]); x2[j] = rT<T>(x2[j]); }
        } else {
            const T* p = (const T*)a.qkv + (long)tok * a.ld + hh * HD + u * 8;
            const V8 t1 = *(const V8*)p, t2 = *(const V8*)(p + HALF);
#pragma unroll
            for (int j = 0; j < 8; ++j) { x1[j] = (float)t1[j]; x2[j] = (float)t2[j]; }
        }
        V8 o1, o2;
        if (hh < a.Hq + a.Hkv) {
            const float* c = a.cs + (long)pos * HD + u * 8;
            const float* s = c + HALF;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o1[j] = (T)(rT<T>(x1[j] * c[j]) + rT<T>(-x2[j] * s[j]));
                o2[j] = (T)(rT<T>(x2[j] * c[j]) + rT<T>(x1[j] * s[j]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { o1[j] = (T)x1[j]; o2[j] = (T)x2[j]; }
        }
        if (hh < a.Hq) {
            T* q = (T*)a.q_out + (long)tok * a.Hq * HD + hh * HD + u * 8;
            *(V8*)q = o1; *(V8*)(q + HALF) = o2;
        } else if (hh < a.Hq + a.Hkv) {
            T* k = (T*)a.Kc + (((long)b * a.Hkv + (hh - a.Hq)) * a.ctx_max + pos) * HD + u * 8;
            *(V8*)k = o1; *(V8*)(k + HALF) = o2;
        } else {
            const int kvh = hh - a.Hq - a.Hkv;
            T* v = (T*)a.Vc + (((long)b * a.Hkv + kvh) * a.ctx_max + pos) * HD + u * 8;
            *(V8*)v = o1; *(V8*)(v + HALF) = o2;
            if (a.Vt) {
                T* vt = (T*)a.Vt + ((long)b * a.Hkv + kvh) * HD * a.vt_ld + pos;
#pragma unroll
                for (int j = 0; j < 8; ++j) { vt[(long)(u * 8 + j) * a.vt_ld] = o1[j]; vt[(long)(HALF + u * 8 + j) * a.vt_ld] = o2[j]; }
            }
        }
    }
}

// Prefill form (round 5): one block per (tile of 16 positions, sequence).  Same arithmetic per element as rope_append_kernel<T, false>; what changes
// is the V^T scratch: the tile's V rows go through LDS and leave as [hd][16 positions] = two 16-byte stores per row, where the per-token kernel
// wrote every V element with its own 2-byte store (8448 tokens x 512 elements: 103 us per layer, 2.9 ms of a 24 ms prefill).
template <typename T>
__global__ __launch_bounds__(256) void rope_append_pf_kernel(RopeAppendArgs a) {
    typedef typename ET<T>::v8 V8;
    constexpr int HD = 128, HALF = 64, TT = 16;
    extern __shared__ __attribute__((aligned(16))) char smem_ra[];            // [TT][Hkv * HD] V values of the tile
    T* sv = (T*)smem_ra;
    const int b = blockIdx.y, p0 = blockIdx.x * TT;
    const int P = a.q_len[b];
    if (p0 >= P) return;
    const int nt = min(TT, P - p0), tok0 = a.q_off[b] + p0;
    const int heads = a.Hq + 2 * a.Hkv, VW = a.Hkv * HD;
    for (int w = threadIdx.x; w < nt * heads * 8; w += blockDim.x) {
        const int t = w / (heads * 8), r = w - t * heads * 8, hh = r >> 3, u = r & 7;
        const int tok = tok0 + t, pos = a.tok_pos[tok];
        const T* p = (const T*)a.qkv + (long)tok * a.ld + hh * HD + u * 8;
        const V8 t1 = *(const V8*)p, t2 = *(const V8*)(p + HALF);
        V8 o1, o2;
        if (hh < a.Hq + a.Hkv) {
            const float* c = a.cs + (long)pos * HD + u * 8;
            const float* s = c + HALF;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x1 = (float)t1[j], x2 = (float)t2[j];
                o1[j] = (T)(rT<T>(x1 * c[j]) + rT<T>(-x2 * s[j]));
                o2[j] = (T)(rT<T>(x2 * c[j]) + rT<T>(x1 * s[j]));
            }
        } else { o1 = t1; o2 = t2; }
        if (hh < a.Hq) {
            T* q = (T*)a.q_out + (long)tok * a.Hq * HD + hh * HD + u * 8;
            *(V8*)q = o1; *(V8*)(q + HALF) = o2;
        } else if (hh < a.Hq + a.Hkv) {
            T* k = (T*)a.Kc + (((long)b * a.Hkv + (hh - a.Hq)) * a.ctx_max + pos) * HD + u * 8;
            *(V8*)k = o1; *(V8*)(k + HALF) = o2;
        } else {
            const int kvh = hh - a.Hq - a.Hkv;
            T* v = (T*)a.Vc + (((long)b * a.Hkv + kvh) * a.ctx_max + pos) * HD + u * 8;
            *(V8*)v = o1; *(V8*)(v + HALF) = o2;
            if (a.Vt) { *(V8*)(sv + t * VW + kvh * HD + u * 8) = o1; *(V8*)(sv + t * VW + kvh * HD + HALF + u * 8) = o2; }
        }
    }
    if (!a.Vt) return;
    __syncthreads();
    // V^T[b][kvh][hd][pos]: row (kvh, hd) of the tile = TT consecutive positions starting at the tile's first position (prompt positions are 0 .. P-1,
    // so p0 is a multiple of 16: the row piece is 32 bytes, 16-byte aligned whenever vt_ld is a multiple of 8)
    const int pos0 = a.tok_pos[tok0];
    for (int rr = threadIdx.x; rr < VW; rr += blockDim.x) {
        T* vt = (T*)a.Vt + ((long)b * a.Hkv * HD + rr) * a.vt_ld + pos0;
        if (nt == TT && (pos0 & 7) == 0 && (a.vt_ld & 7) == 0) {
            V8 lo, hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) { lo[j] = sv[j * VW + rr]; hi[j] = sv[(8 + j) * VW + rr]; }
            *(V8*)vt = lo; *(V8*)(vt + 8) = hi;
        } else {
            for (int j = 0; j < nt; ++j) vt[j] = sv[j * VW + rr];
        }
    }
}

// ---------------------------------------------------------------- embedding gather / audio scatter (modeling_glmasr.py:452-465)
// src[tok] >= 0: row of the embedding table; src[tok] < 0: audio row -(src+1) of `audio`.  (2-byte copies: dtype-agnostic)
__global__ void assemble_embeds_kernel(const int* src, const bf16_t* table, const bf16_t* audio, bf16_t* x, int n_tok, int d) {
    const int tok = blockIdx.x;
    const int s = src[tok];
    const bf16_t* from = s >= 0 ? table + (long)s * d : audio + (long)(-(s + 1)) * d;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) *(bf16x8*)(x + (long)tok * d + c * 8) = *(const bf16x8*)(from + c * 8);
}

// ---------------------------------------------------------------- argmax + greedy controller (generation/utils.py:2894-2936)
template <typename T>
__global__ __launch_bounds__(1024) void greedy_kernel(GreedyArgs a) {
    typedef typename ET<T>::v8 V8;
    __shared__ float sv[16];
    __shared__ int si[17];
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* lg = a.logits + (long)b * a.V;
    const long ks_stride = (long)a.mpad * a.V;
    float* dump = a.logits_dump ? a.logits_dump + (long)a.step_counter[b] * a.dump_stride_step + (long)b * a.V : nullptr;
    float best = -INFINITY; int bi = 0x7fffffff;
    // the first layer's norm weight for the tail of this kernel, requested before anything else (cold every step; behind the token's
    // embedding row it was one more dependent round trip)
    f32x4 gw0 = {0.f, 0.f, 0.f, 0.f}, gw1 = {0.f, 0.f, 0.f, 0.f};
    if (a.y && (a.d >> 3) <= 1024 && tid < (a.d >> 3)) { gw0 = *(const f32x4*)(a.norm_w + tid * 8); gw1 = *(const f32x4*)(a.norm_w + tid * 8 + 4); }
    // four strides per trip with all their slab loads issued first (the rolled form paid one L2 round trip per stride)
    constexpr int U = 4;
    for (int i0 = tid * 4; i0 < a.V; i0 += 1024 * 4 * U) {
        f32x4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 4096, ic = i < a.V ? i : 0;
            v[u] = *(const f32x4*)(lg + ic);
            w[u] = *(const f32x4*)(lg + (a.ksplit > 1 ? ks_stride : 0) + ic);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 4096;
            if (i >= a.V) break;
            f32x4 t = v[u];
            if (a.ksplit > 1) t += w[u];
            for (int ks = 2; ks < a.ksplit; ++ks) t += *(const f32x4*)(lg + ks * ks_stride + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = rT<T>(t[j]);        // logits are T in the reference, compared as fp32
                if (dump) dump[i + j] = r;
                if (r > best) { best = r; bi = i + j; }   // strict > keeps the first maximum within a thread
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { sv[wid] = best; si[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        int tok = bi;
        const int fin = a.finished[b];
        bool running = false;
        if (fin) tok = a.pad_id;                 // finished rows emit the pad token (:2928-2929)
        else {
            if (a.force_ids) tok = a.force_ids[(long)b * a.force_ld + a.n_new[b]];   // teacher forcing (parity tests): feed this id instead
            a.out_ids[(long)b * a.out_ld + a.n_new[b]] = tok;
            const int nn = a.n_new[b] + 1;
            a.n_new[b] = nn;
            bool stop = nn >= a.max_new[b];
            for (int e = 0; e < a.n_eos; ++e) stop |= (tok == a.eos[e]);
            if (stop) { a.finished[b] = 1; atomicSub(a.n_active, 1); } else running = true;
        }
        // device error word (SkinnyArgs.err = n_active[1]): a kernel of this step gave up on an in-kernel wait, its outputs are garbage.  The count of
        // running rows goes (and stays) far below zero: the loop stops at its next check and the host fails the batch (DEV_ERR_ACTIVE in engine.cpp)
        if (a.dev_err && b == 0 && *a.dev_err) atomicMin(a.n_active, -(1 << 24));
        // Only a row that keeps running advances its context.  A finished row stays where it is (its later steps rewrite the same
        // cache slot and are discarded), so kv_len never exceeds prompt + max_new - 1 < max_ctx whatever the other rows' budgets are:
        // before, a [long prompt, small budget] row riding a [short prompt, large budget] batch walked past its cache region.
        if (running) {
            a.tok_pos[b] = a.kv_len[b];          // the new token sits right after the current context
            a.kv_len[b] += 1;
        }
        s_tok = tok;
        if (a.step_counter) a.step_counter[b] += 1;
    }
    __syncthreads();
    const T* row = (const T*)a.table + (long)s_tok * a.d;
    T* xo = (T*)a.x;
    if (!a.y || (a.d >> 3) > 1024) {
        for (int c = tid; c < (a.d >> 3); c += 1024) *(V8*)(xo + (long)b * a.d + c * 8) = *(const V8*)(row + c * 8);
        return;
    }
    // next step's input row and, in the same pass, the first decoder layer's input RMSNorm of it (modeling_llama.py:60-65, :306):
    // one launch less per token step
    const int c = tid, nv = a.d >> 3;
    V8 xv;
    float ss = 0.f;
    if (c < nv) {
        xv = *(const V8*)(row + c * 8);
        *(V8*)(xo + (long)b * a.d + c * 8) = xv;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)xv[j]; ss += f * f; }
    }
    ss = wave_sum(ss);
    __syncthreads();                         // sv is reused below
    if (lane == 0) sv[wid] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += sv[w];
    const float r = 1.0f / sqrtf(tot / a.d + a.norm_eps);
    float yo[8];
    if (c < nv) {
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = (T)((j < 4 ? gw0[j & 3] : gw1[j & 3]) * rT<T>((float)xv[j] * r)); yo[j] = (float)o[j]; }
        *(V8*)((T*)a.y + (long)b * a.d + c * 8) = o;
    }
    if (a.qo.q) quant_emit_row(yo, c < nv, c, b, a.qo, sv, si);
}

// ---------------------------------------------------------------- misc
// In-stream fill of small control words.  hipMemsetAsync of a few bytes was observed not to be reliably ordered against
// the neighbouring kernels of a non-blocking stream (a stale per-segment log-mel maximum survived a reset on some runs);
// a kernel on the stream is.
__global__ void fill_i32_kernel(int* p, int value, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = value;
}
void launch_fill_i32(int* p, int value, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, value, n);
}
template <typename T> __global__ void f32_to_t_kernel(const float* in, T* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)in[i];
}
template <typename T> __global__ void t_to_f32_kernel(const T* in, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
__global__ void bf16_to_f16_kernel(const bf16_t* in, f16_t* out, long n) {   // a bf16 checkpoint loaded with torch_dtype=float16 (asr.py:156)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (f16_t)(float)in[i];
}

// sonicscribe_amd/synth.py restated for the device: writes bf16 and/or fp32
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__global__ void synth_fill_kernel(unsigned long long key, long n, float scale, float offset, bf16_t* out_bf, float* out_f32, int round_f32) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long z = mix64(key + (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ULL);
    const int bits = (int)(z >> 40);
    const float r = __fsub_rn(__fmul_rn((float)bits, 0x1p-23f), 1.0f);
    const float v = __fadd_rn(offset, __fmul_rn(r, scale));
    if (out_bf) out_bf[i] = f2bf(v);
    if (out_f32) out_f32[i] = (out_bf || round_f32) ? rbf(v) : v;      // fp32 alone: the generator's exact value (synth.synth_fill(..., bf16=False)) unless round_f32
}

// ---------------------------------------------------------------- launchers
void launch_layernorm(const bf16_t* x, const float* w, const float* b, bf16_t* y, int rows, int d, float eps, hipStream_t s, int dt, const QuantActArgs* qa) {
    const QuantActArgs q = qa ? *qa : QuantActArgs{};
    if (q.q) { DT_SWITCH(dt, T, hipLaunchKernelGGL((layernorm_kernel<T, true>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, w, b, (T*)y, rows, d, eps, q)); }
    else { DT_SWITCH(dt, T, hipLaunchKernelGGL((layernorm_kernel<T, false>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, w, b, (T*)y, rows, d, eps, q)); }
}
void launch_rmsnorm(const bf16_t* x, const float* w, bf16_t* y, int rows, int d, float eps, const int* row_map, hipStream_t s, int dt, const QuantActArgs* qa) {
    const QuantActArgs q = qa ? *qa : QuantActArgs{};
    if (q.q) { DT_SWITCH(dt, T, hipLaunchKernelGGL((rmsnorm_kernel<T, true>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, w, (T*)y, rows, d, eps, row_map, q)); }
    else { DT_SWITCH(dt, T, hipLaunchKernelGGL((rmsnorm_kernel<T, false>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, w, (T*)y, rows, d, eps, row_map, q)); }
}
void launch_add_rmsnorm(bf16_t* x, const float* P, int ksplit, int mpad, const float* w, bf16_t* y, int rows, int d, float eps, hipStream_t s,
                        int dt, const DeqInfo* dq, const QuantOut* qo, const PrefetchRange* pf, int pf_blocks) {
    const int threads = ((d >> 3) + 63) / 64 * 64;   // d <= 2048 -> <= 256 threads
    const DeqInfo q = dq ? *dq : DeqInfo{};
    const QuantOut o = qo ? *qo : QuantOut{};
    const PrefetchRange r = pf ? *pf : PrefetchRange{nullptr, 0};
    const int extra = pf && pf->p && pf_blocks > 0 ? pf_blocks : 0;
    DT_SWITCH(dt, T, hipLaunchKernelGGL(add_rmsnorm_kernel<T>, dim3(rows + extra), dim3(threads), 0, s, (T*)x, P, ksplit, mpad, w, (T*)y, rows, d, eps, q, o, r));
}
void launch_swiglu_slab(const float* P, int ksplit, int mpad, int n2, bf16_t* act, int rows, hipStream_t s, int dt, int gu8) {
    const long n = (long)rows * (n2 >> 3);
    DT_SWITCH(dt, T, hipLaunchKernelGGL(swiglu_slab_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s, P, ksplit, mpad, n2, (T*)act, rows, gu8));
}
void launch_swiglu_quant(const float* P, int ksplit, int mpad, int ff, bf16_t* act, int rows, const DeqInfo& dq, const QuantOut& qo, hipStream_t s) {
    hipLaunchKernelGGL(swiglu_quant_kernel, dim3(rows), dim3(1024), 0, s, P, ksplit, mpad, ff, (f16_t*)act, dq, qo);
}
void launch_quant_rows(const bf16_t* X, long ld, int M, int K, const QuantOut& qo, hipStream_t s) {
    hipLaunchKernelGGL(quant_rows_kernel, dim3(M), dim3(1024), 0, s, (const f16_t*)X, ld, K, qo);
}
void launch_rope_enc(bf16_t* qk, long ld, int M, int T, int heads2, int hd, int rd, const float* cs, hipStream_t s, int dt) {
    const long n = (long)M * heads2 * (rd >> 4);
    DT_SWITCH(dt, E, hipLaunchKernelGGL(rope_enc_kernel<E>, dim3((n + 255) / 256), dim3(256), 0, s, (E*)qk, ld, M, T, heads2, hd, rd, cs));
}
void launch_rope_append(const RopeAppendArgs& a, bool slab, hipStream_t s) {
    if (a.n_tok <= 0) return;
    if (!slab && a.q_off && a.q_len && a.n_seq > 0 && a.max_p > 0 && a.Hkv * 128 * 16 * 2 <= 65536) {      // prefill: tiles of 16 positions per sequence
        DT_SWITCH(a.dt, T, hipLaunchKernelGGL((rope_append_pf_kernel<T>), dim3((a.max_p + 15) / 16, a.n_seq), dim3(256), (size_t)a.Hkv * 128 * 16 * 2, s, a));
        return;
    }
    DT_SWITCH(a.dt, T, {
        if (slab) hipLaunchKernelGGL((rope_append_kernel<T, true>), dim3(a.n_tok), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((rope_append_kernel<T, false>), dim3(a.n_tok), dim3(256), 0, s, a);
    });
}
void launch_assemble_embeds(const int* src, const bf16_t* table, const bf16_t* audio, bf16_t* x, int n_tok, int d, hipStream_t s) {
    if (n_tok > 0) hipLaunchKernelGGL(assemble_embeds_kernel, dim3(n_tok), dim3(256), 0, s, src, table, audio, x, n_tok, d);
}
void launch_greedy(const GreedyArgs& a, hipStream_t s) {
    if (a.dt == DT_F32) { hipLaunchKernelGGL(greedy_kernel<float>, dim3(a.B), dim3(1024), 0, s, a); return; }   // SONIC_MODE_F32: fp32 logits, table and rows
    DT_SWITCH(a.dt, T, hipLaunchKernelGGL(greedy_kernel<T>, dim3(a.B), dim3(1024), 0, s, a));
}
void launch_f32_to_bf16(const float* in, bf16_t* out, long n, hipStream_t s, int dt) {
    if (n > 0) DT_SWITCH(dt, T, hipLaunchKernelGGL(f32_to_t_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s, in, (T*)out, n));
}
void launch_bf16_to_f32(const bf16_t* in, float* out, long n, hipStream_t s, int dt) {
    if (n > 0) DT_SWITCH(dt, T, hipLaunchKernelGGL(t_to_f32_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s, (const T*)in, out, n));
}
void launch_bf16_to_f16(const bf16_t* in, bf16_t* out, long n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(bf16_to_f16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, (f16_t*)out, n);
}
void launch_synth_fill(unsigned long long key, long n, float scale, float offset, bf16_t* out_bf, float* out_f32, hipStream_t s, int round_f32) {
    if (n > 0) hipLaunchKernelGGL(synth_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, key, n, scale, offset, out_bf, out_f32, round_f32);
}
