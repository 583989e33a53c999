// Attention kernels for gfx950.
//
// flash_attn_kernel<HD, CAUSAL>: one pass, online softmax, bf16 MFMA (v_mfma_f32_16x16x32_bf16).
//   encoder: HD=64, non-causal, T=1500 keys, 20 heads          (modeling_glmasr.py:171-221, mask=None)
//   prefill: HD=128, causal, GQA 16:4, ragged prompt lengths    (modeling_llama.py:217-281 + sdpa is_causal)
// Formulation: S^T = K.Q^T and O^T = V^T.P^T, i.e. the QUERY sits on the MFMA column (lane & 15):
//   * every accumulator register of a lane belongs to one query -> softmax statistics, the rescale
//     factor and the final 1/l are per-lane scalars, no cross-lane traffic inside the tile loop
//     except one 2-step max exchange between the four 16-lane groups;
//   * the S^T accumulator, converted to bf16, *is* the B-operand fragment of the P.V product
//     (keys on k): K rows are fed in the order key = 8*(r>>2) + 4*b + (r&3) so that a lane's 8
//     accumulators of a 32-key step are the 8 consecutive keys 8*g .. 8*g+7 its fragment needs;
//   * V is consumed transposed (V^T[hd][key], written that way by the QKV GEMM epilogue / the
//     rope+append kernel), so both LDS operands are read as single 16-byte fragments.
// LDS tiles are XOR-swizzled so that every ds_read_b128 lane group touches 16 distinct 16-B slots.
//
// decode_attn_kernel: q_len = 1 over the KV cache, HBM-bound, coalesced 1 KiB wave loads of K and V
// rows, fp32 scores in LDS, one block per (sequence, kv head) serving its 4 query heads together.
#include "common.h"
#include "kernels.h"
#include "int8_util.h"
#include <type_traits>


// reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with gfx950's permlane swaps: VALU only, no LDS round
// trip (ds_bpermute).  permlane16_swap exchanges the odd rows of one operand with the even rows of the other, permlane32_swap the
// upper half of one with the lower half of the other; with both operands equal the two results hold the two partners in every lane.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows_sum(float v) {
    u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows_max(float v) {
    u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int HD> __device__ __forceinline__ int kswz(int row) {  // swizzle term for the K tile (rows = keys)
    if (HD == 64) return ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
    return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
}

// VAR (A/B, option "flash_variant"): bit 0 = two LDS tile buffers, one barrier per key tile; bit 1 = accumulators rescaled only when a
// running maximum of the wave moved
template <typename T, int HD, bool CAUSAL, int VAR>
__global__ __launch_bounds__(256) void flash_attn_kernel(FlashArgs a) {
    constexpr bool DB = (VAR & 1) != 0, LAZY = (VAR & 2) != 0;
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    constexpr int HS = HD / 32;        // hd k-steps for S^T
    constexpr int HB = HD / 16;        // hd blocks of O^T
    constexpr int KCH = HD / 8;        // 16-B chunks per K row
    constexpr int KROW = HD * 2;       // bytes per K row
    constexpr int KT_BYTES = 64 * KROW;
    constexpr int VT_BYTES = HD * 128;
    constexpr int KV_PASSES = KT_BYTES / 4096;  // 256 threads x 16 B per pass
    // two K / V^T tile buffers: tile kt+1 is committed to the other buffer while tile kt is being read, one barrier per tile
    __shared__ __attribute__((aligned(16))) char smem[(DB ? 2 : 1) * (KT_BYTES + VT_BYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, kh = h / (a.Hq / a.Hkv);
    const int q_len = a.q_len ? a.q_len[b] : a.T;
    const int kv_len = a.kv_len ? a.kv_len[b] : a.T;
    const int q0 = blockIdx.x * 128;
    if (q0 >= q_len) return;
    const long qbase = a.q_off ? (long)a.q_off[b] * a.q_ld : (long)b * a.q_seq_stride;
    const long obase = a.q_off ? (long)a.q_off[b] * a.o_ld : (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld;
    const T* Q = (const T*)a.Q + qbase + (long)h * HD;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)kh * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)kh * a.vt_head_stride;
    const int qpos_off = kv_len - q_len;  // absolute position of query 0 (causal)

    // Q fragments (B-operand): lane holds Q[query = qb*16 + fr][hd = hs*32 + fg*8 .. +7]
    V8 qf[2][HS];
    int qrow[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = q0 + wid * 32 + qb * 16 + fr;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) qf[qb][hs] = *(const V8*)(Q + (long)qr * a.q_ld + hs * 32 + fg * 8);
    }

    // Make the compiler retire the Q loads HERE: otherwise its waitcnt bookkeeping carries them into the tile loop as "possibly
    // pending" and every iteration waits for vmcnt(0) in front of the first MFMAs - i.e. for the K/V prefetch it has just issued.
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) asm volatile("" : "+v"(qf[qb][hs]));
    f32x4 oacc[2][HB];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f}, lrun[2] = {0.f, 0.f};

    int n_tiles = (kv_len + 63) / 64;
    if (CAUSAL) {
        const int last_q = min(q0 + 127, q_len - 1) + qpos_off;
        n_tiles = min(n_tiles, last_q / 64 + 1);
    }

    // register-staged prefetch (issue-early / write-late)
    V8 kreg[KV_PASSES], vreg[KV_PASSES];
    auto issue = [&](int kt) {
        const int key0 = kt * 64;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid;
            const int kr = idx / KCH, kc = idx % KCH;        // K tile: row = key, chunk of 8 hd
            kreg[p] = *(const V8*)(K + (long)(key0 + kr) * a.k_ld + kc * 8);
            const int vr = idx >> 3, vc = idx & 7;           // V^T tile: row = hd, chunk of 8 keys
            vreg[p] = *(const V8*)(Vt + (long)vr * a.vt_ld + key0 + vc * 8);
        }
    };
    auto commit = [&](int buf) {
        char* sK = smem + buf * (KT_BYTES + VT_BYTES);
        char* sV = sK + KT_BYTES;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid;
            const int kr = idx / KCH, kc = idx % KCH;
            *(V8*)(sK + kr * KROW + ((kc ^ kswz<HD>(kr)) << 4)) = kreg[p];
            const int vr = idx >> 3, vc = idx & 7;
            *(V8*)(sV + vr * 128 + ((vc ^ (vr & 7)) << 4)) = vreg[p];
        }
    };

    // Round 2 had one buffer and two barriers per tile (consumed -> commit -> visible).  With two buffers the registers that hold tile
    // kt+1 are committed to the buffer tile kt-1 was read from (every wave passed the barrier that closed iteration kt-1), tile kt+2 is
    // requested, tile kt is consumed, and ONE barrier closes the iteration.
    issue(0);
    if (DB) {
        commit(0);
        if (n_tiles > 1) issue(1);
        __syncthreads();
    }
    for (int kt = 0; kt < n_tiles; ++kt) {
        if (DB) {
            if (kt + 1 < n_tiles) { commit((kt + 1) & 1); if (kt + 2 < n_tiles) issue(kt + 2); }
        } else {
            __syncthreads();          // previous tile fully consumed
            commit(0);
            __syncthreads();
            if (kt + 1 < n_tiles) issue(kt + 1);
        }
        const char* sK = smem + (DB ? (kt & 1) : 0) * (KT_BYTES + VT_BYTES);
        const char* sV = sK + KT_BYTES;
        const int key0 = kt * 64;

        // ---- S^T = K . Q^T   (st[ks][sb][qb][j]: key = key0 + ks*32 + 8*fg + 4*sb + j, query = qb*16 + fr)
        f32x4 st[2][2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int krow = ks * 32 + 8 * (fr >> 2) + 4 * sb + (fr & 3);
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hs = 0; hs < HS; ++hs) {
                    const int c = hs * 4 + fg;
                    const V8 kf = *(const V8*)(sK + krow * KROW + ((c ^ kswz<HD>(krow)) << 4));
                    s0 = ET<T>::mfma(kf, qf[0][hs], s0);
                    s1 = ET<T>::mfma(kf, qf[1][hs], s1);
                }
                st[ks][sb][0] = s0;
                st[ks][sb][1] = s1;
            }

        // ---- online softmax per query (= per lane), P^T fragments.  Statistics are kept on the RAW scores (scale > 0, so the max
        // commutes with the scaling) and the scale is folded into the exponent: p = exp2(s * c - m * c), c = scale * log2(e), one FMA
        // and one v_exp_f32 per score.  Tiles that need no masking (all but the last key tile / the causal diagonal) take a branch
        // without the per-element compares and selects.
        const bool edge = (key0 + 64 > kv_len) || (CAUSAL && (key0 + 63 > q0 + wid * 32 + qpos_off));
        V8 pf[2][2];
        const float cexp = a.scale * 1.44269504088896341f;
        float alph[2];
        auto softmax = [&](auto masked) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (decltype(masked)::value) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int key = key0 + ks * 32 + 8 * fg + 4 * sb + j;
                                bool ok = key < kv_len;
                                if (CAUSAL) ok = ok && (key <= qrow[qb] + qpos_off);
                                st[ks][sb][qb][j] = ok ? st[ks][sb][qb][j] : -1e30f;
                            }
                }
                float mx = -1e30f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, st[ks][sb][qb][j]);
                mx = rows_max(mx);
                const float mnew = fmaxf(mrun[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * cexp);
                mrun[qb] = mnew;
                const float moff = -mnew * cexp;
                float psum = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[ks][sb][qb][j], cexp, moff));
                            psum += p;
                            pf[qb][ks][sb * 4 + j] = (T)p;
                        }
                lrun[qb] = lrun[qb] * alpha + psum;
                alph[qb] = alpha;
            }
        };
        if (edge) softmax(std::true_type{}); else softmax(std::false_type{});
        // the accumulators are rescaled only when some query of this wave moved its running maximum (alpha == 1 otherwise: after the first
        // tiles that is the common case, and the 8 * HB multiplies are pure VALU time in a VALU-bound loop); one wave-uniform branch
        if (!LAZY || __builtin_amdgcn_ballot_w64(alph[0] != 1.0f || alph[1] != 1.0f) != 0) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] *= alph[qb];
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const int vr = hb * 16 + fr, c = ks * 4 + fg;
                const V8 vf = *(const V8*)(sV + vr * 128 + ((c ^ (vr & 7)) << 4));
                oacc[0][hb] = ET<T>::mfma(vf, pf[0][ks], oacc[0][hb]);
                oacc[1][hb] = ET<T>::mfma(vf, pf[1][ks], oacc[1][hb]);
            }
        if (DB) __syncthreads();  // tile kt consumed by every wave; tile kt+1's commit visible
    }

    // ---- epilogue: O[query][h*HD + hb*16 + fg*4 + j] = O^T / l
    T* O = (T*)a.O + obase + (long)h * HD;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = rows_sum(lrun[qb]);
        if (qrow[qb] < q_len) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (T)(oacc[qb][hb][j] / l);
                *(V4*)(O + (long)qrow[qb] * a.o_ld + hb * 16 + fg * 4) = o;
            }
        }
    }
}

template <int VAR> static void launch_flash_v(const FlashArgs& a, int hd, bool causal, dim3 grid, hipStream_t s) {
    dim3 block(256);
    DT_SWITCH(a.dt, T, {
        if (hd == 64 && !causal) hipLaunchKernelGGL((flash_attn_kernel<T, 64, false, VAR>), grid, block, 0, s, a);
        else if (hd == 64 && causal) hipLaunchKernelGGL((flash_attn_kernel<T, 64, true, VAR>), grid, block, 0, s, a);
        else if (hd == 128 && causal) hipLaunchKernelGGL((flash_attn_kernel<T, 128, true, VAR>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((flash_attn_kernel<T, 128, false, VAR>), grid, block, 0, s, a);
    });
}
void launch_flash(const FlashArgs& a, int hd, bool causal, int B, int max_q, hipStream_t s) {
    // the encoder's shape (head dim 64, every key visible) has its own kernel since round 5 (attn_enc.hip: 32x32x16 MFMAs, 64 queries per wave,
    // LDS-DMA tiles, running maximum fixed after the first key tile)
    if (hd == 64 && !causal && g_opts.flash_enc > 0 && a.Hq == a.Hkv) { launch_flash_enc(a, B, max_q, g_opts.flash_enc - 1, s); return; }
    dim3 grid((max_q + 127) / 128, a.Hq, B);
    switch (g_opts.flash_variant & 3) {
        case 1: launch_flash_v<1>(a, hd, causal, grid, s); break;
        case 2: launch_flash_v<2>(a, hd, causal, grid, s); break;
        case 3: launch_flash_v<3>(a, hd, causal, grid, s); break;
        default: launch_flash_v<0>(a, hd, causal, grid, s); break;
    }
}

// ------------------------------------------------------------------------------------------------

#ifdef SONIC_AB      // A/B builds only (make SONIC_AB=1): the product library carries what runs
// Round-2 form of the decode attention (P.V on the VALU), kept for A/B runs (option "decode_attn_v1").
// HD = 128, group size G = Hq/Hkv <= 4.  One block (8 waves) per (sequence, kv head).
//  0. the first 16-key slice of K and V of every wave is requested before anything else (addresses do not depend on kv_len:
//     rows are clamped to the cache and masked later), so its HBM latency overlaps the prologue;
//  1. prologue (fused RoPE + KV append, modeling_llama.py:121-143,261-262): sums the QKV skinny-GEMM slabs of this
//     (sequence, kv head), rounds to bf16, applies rotate-half RoPE at the token's position, writes the new K / V rows
//     into the cache and keeps q (4 heads), k, v in LDS;
//  2. single pass over the cached keys, 16 keys per wave-iteration (next slice prefetched into registers):
//     scores S^T[key][head] = K.q^T with 4 v_mfma_f32_16x16x32_bf16 (a K row piece per lane IS the A fragment; the q heads sit
//     on 4 of the 16 B columns), per-wave online softmax, probabilities handed to the lanes that own the V pieces through a
//     256-byte per-wave LDS record, P.V on the VALU (V is row-major in the cache);
//  3. the new key comes from LDS; the 8 waves' (m, l, acc) states are merged through LDS.
// Probabilities are rounded to bf16 for the P.V product and kept in fp32 for the row sum, as in the prefill kernel.
template <typename T>
__global__ __launch_bounds__(512) void decode_attn_v1_kernel(DecodeAttnArgs a) {
    typedef typename ET<T>::v8 V8;
    constexpr int HD = 128, HALF = 64, GMAX = 4, NW = 8;
    __shared__ float s_acc[NW][GMAX][HD];
    __shared__ float s_m[NW][GMAX], s_l[NW][GMAX];
    __shared__ __attribute__((aligned(16))) T s_q[GMAX + 2][HD];   // q heads (unscaled), then k, v of the new token (all T values)
    __shared__ __attribute__((aligned(16))) float s_p[NW][16][4];  // per wave: probabilities [key in slice][head]
    __shared__ __attribute__((aligned(16))) float s_mn[NW][4];     // per wave: new running max per head
    const int G = a.Hq / a.Hkv;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;                        // MFMA: A row / k-chunk;  V pieces: hd chunk r of key 4u + g
    const int b = blockIdx.x, kvh = blockIdx.y;
    T* Kc = (T*)a.Kc + ((long)b * a.Hkv + kvh) * a.ctx_max * HD;
    T* Vc = (T*)a.Vc + ((long)b * a.Hkv + kvh) * a.ctx_max * HD;
    const int cm1 = a.ctx_max - 1;

    KT(a, 0);
    V8 kf[4], vv[4], kfn[4], vvn[4];
    auto load = [&](int k0, V8 (&kd)[4], V8 (&vd)[4]) {
        const int kr = min(k0 + r, cm1);
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) kd[hs] = *(const V8*)(Kc + (long)kr * HD + hs * 32 + g * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) vd[u] = *(const V8*)(Vc + (long)min(k0 + 4 * u + g, cm1) * HD + r * 8);
    };
    load(wid * 16, kf, vv);

    int n;                                   // keys visible, the new token included at position n-1
    if (a.P) {
        // QKV slab sum of this (segment, kv head): issued before kv_len is needed, so the two latencies overlap
        const int N = (a.Hq + 2 * a.Hkv) * HD;
        const int w = tid, vi = w / HALF, i = w % HALF;      // vector (q heads.., k, v), index in the first half; (G + 2) * 64 <= 384 threads
        const bool act = w < (G + 2) * HALF;
        float x1 = 0.f, x2 = 0.f;
        if (act) {
            const int col = (vi < G ? (kvh * G + vi) : vi == G ? (a.Hq + kvh) : (a.Hq + a.Hkv + kvh)) * HD + i;
            if (a.dq.sca) {
                // int8 mode: int32 slabs of the quantised q/k/v projections -> fp16 module outputs (LLM.int8 dequant + outliers)
                x1 = deq1(a.dq, a.P, a.ksplit, a.mpad, b, col, N); x2 = deq1(a.dq, a.P, a.ksplit, a.mpad, b, col + HALF, N);
            } else {
                // all slab loads in flight at once (a rolled loop would serialise one L2 round trip per slab); fixed summation order
                float v1[8], v2[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const float* p = a.P + ((long)(ks < a.ksplit ? ks : 0) * a.mpad + b) * N + col;
                    v1[ks] = p[0]; v2[ks] = p[HALF];
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    if (ks < a.ksplit) { x1 += v1[ks]; x2 += v2[ks]; }
            }
        }
        n = min(max(a.kv_len[b], 1), a.ctx_max);      // (clamped: the cache / RoPE table rows of this block end at ctx_max)
        if (act) {
            const int pos = n - 1;
            x1 = rT<T>(x1); x2 = rT<T>(x2);
            float o1 = x1, o2 = x2;
            if (vi <= G) {
                const float c = a.cs[(long)pos * HD + i], sn = a.cs[(long)pos * HD + HALF + i];
                o1 = rT<T>(rT<T>(x1 * c) + rT<T>(-x2 * sn));
                o2 = rT<T>(rT<T>(x2 * c) + rT<T>(x1 * sn));
            }
            const int row = vi < G ? vi : (vi == G ? GMAX : GMAX + 1);
            const T b1 = (T)o1, b2 = (T)o2;
            s_q[row][i] = b1; s_q[row][HALF + i] = b2;
            if (vi == G) { Kc[(long)pos * HD + i] = b1; Kc[(long)pos * HD + HALF + i] = b2; }
            if (vi == G + 1) { Vc[(long)pos * HD + i] = b1; Vc[(long)pos * HD + HALF + i] = b2; }
        }
        __syncthreads();
    } else {
        n = min(max(a.kv_len[b], 1), a.ctx_max);
    }
    KT(a, 1);
    // q as the MFMA B operand: column r = head (zero beyond the group), k = head-dim
    V8 qf[4];
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) {
        const V8 t = a.P ? *(const V8*)&s_q[r < G ? r : 0][hs * 32 + g * 8]
                         : *(const V8*)((const T*)a.Q + (long)b * a.Hq * HD + (kvh * G + (r < G ? r : 0)) * HD + hs * 32 + g * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[hs][j] = r < G ? t[j] : (T)0.f;
    }
    float m[GMAX], lsum = 0.f, acc[GMAX][8];
#pragma unroll
    for (int h = 0; h < GMAX; ++h) {
        m[h] = -1e30f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[h][i] = 0.f;
    }
    auto step = [&](int k0, const V8 (&kd)[4], const V8 (&vd)[4], int limit) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) st = ET<T>::mfma(kd[hs], qf[hs], st);
        float sc[4], mx = -1e30f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = (k0 + g * 4 + j) < limit ? st[j] * a.scale : -1e30f; mx = fmaxf(mx, sc[j]); }
        mx = rows_max(mx);
        const float m_r = r == 0 ? m[0] : r == 1 ? m[1] : r == 2 ? m[2] : m[3];
        const float mn = fmaxf(m_r, mx);
        float ps = 0.f, pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float pv = (k0 + g * 4 + j) < limit ? __expf(sc[j] - mn) : 0.f; ps += pv; pr[j] = rT<T>(pv); }
        lsum = lsum * __expf(m_r - mn) + ps;
        if (r < 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s_p[wid][g * 4 + j][r] = pr[j];
            if (g == 0) s_mn[wid][r] = mn;
        }
        __builtin_amdgcn_wave_barrier();           // same-wave LDS traffic is in order; this only pins the compiler's schedule
        const f32x4 mn4 = *(const f32x4*)s_mn[wid];
        f32x4 pk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pk[u] = *(const f32x4*)s_p[wid][4 * u + g];
#pragma unroll
        for (int h = 0; h < GMAX; ++h) {
            const float alpha = __expf(m[h] - mn4[h]);
            m[h] = mn4[h];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = acc[h][i] * alpha;
#pragma unroll
                for (int u = 0; u < 4; ++u) t += pk[u][h] * (float)vd[u][i];
                acc[h][i] = t;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    const int nc = a.P ? n - 1 : n;          // keys read from the cache
    for (int k0 = wid * 16; k0 < nc; k0 += NW * 16) {
        const bool more = k0 + NW * 16 < nc;
        if (more) load(k0 + NW * 16, kfn, vvn);
        step(k0, kf, vv, nc);
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { kf[u] = kfn[u]; vv[u] = vvn[u]; }
        }
    }
    if (a.P && wid == NW - 1) {              // the token being decoded: its k / v are still in LDS (slice key 0 only); the last wave has
                                             // the fewest cached slices (slices go round-robin from wave 0), so it takes the extra step
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) kf[hs] = *(const V8*)&s_q[GMAX][hs * 32 + g * 8];
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = *(const V8*)&s_q[GMAX + 1][r * 8];
        step(0, kf, vv, 1);
    }
    KT(a, 2);
    // merge: per wave the row sums over the 4 key quarters, the outputs over the 4 V-owner groups; then the 8 waves
    lsum = rows_sum(lsum);
    if (g == 0 && r < 4) s_l[wid][r] = lsum;
    if (lane == 0) { s_m[wid][0] = m[0]; s_m[wid][1] = m[1]; s_m[wid][2] = m[2]; s_m[wid][3] = m[3]; }
#pragma unroll
    for (int h = 0; h < GMAX; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = rows_sum(acc[h][i]);
            if (g == 0) s_acc[wid][h][r * 8 + i] = v;
        }
    __syncthreads();
    KT(a, 3);
    for (int idx = tid; idx < G * HD; idx += 512) {
        const int h = idx / HD, e = idx % HD;
        float M = -1e30f;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][h]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const float f = __expf(s_m[w][h] - M); num += f * s_acc[w][h][e]; den += f * s_l[w][h]; }
        ((T*)a.O)[(long)b * a.Hq * HD + (kvh * G + h) * HD + e] = (T)(num / den);
    }
    KT(a, 4);
}
#endif

// 16x16x16 MFMA on the engine's 16-bit element type (the P.V product of the decode attention: 16 keys per step)
template <typename T> struct PV16;
typedef short s16x4_t __attribute__((ext_vector_type(4)));
template <> struct PV16<bf16_t> {
    static __device__ __forceinline__ f32x4 mfma(u32x2_t a, u32x2_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
    }
};
template <> struct PV16<f16_t> {
    static __device__ __forceinline__ f32x4 mfma(u32x2_t a, u32x2_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
};

// HD = 128, group size G = Hq/Hkv <= 4.  One block (8 waves) per (sequence, kv head).
//  0. the first 16-key slice of K and V of every wave is requested before anything else (addresses do not depend on kv_len:
//     rows are clamped to the cache and masked later), so its HBM latency overlaps the prologue;
//  1. prologue (fused RoPE + KV append, modeling_llama.py:121-143,261-262): sums the QKV skinny-GEMM slabs of this
//     (sequence, kv head), rounds to bf16, applies rotate-half RoPE at the token's position, writes the new K / V rows
//     into the cache and keeps q (4 heads), k, v in LDS;
//  2. single pass over the cached keys, 16 keys per wave-iteration (next slice prefetched into registers), BOTH products on the
//     matrix pipe:
//       S^T[key][head] = K.q^T   4 x v_mfma_f32_16x16x32 (a K row piece per lane IS the A fragment; the q heads sit on 4 of the 16 B
//                                columns).  D layout: lane (r, g) holds head r, keys 4g .. 4g+3;
//       O[head][hd]   += P.V     8 x v_mfma_f32_16x16x16: the A fragment of lane (r, g) is P[head r][keys 4g .. 4g+3] - exactly the
//                                probabilities that lane just computed, no exchange; the B fragment of tile t is V[keys 4g .. 4g+3][hd
//                                8r + t]: every lane loads the 16-byte piece r of its four key rows (full 256-byte rows per instruction,
//                                as before) and tile t takes element t of each piece (two v_perm_b32 per tile), i.e. tile t covers
//                                the head-dim columns {8n + t}.
//     Round 2 did P.V on the VALU (128 FMAs + 32 converts per lane and slice, probabilities handed over through LDS): the key loop was
//     VALU-bound at 0.9 us per slice (in-kernel timeline, profiles/round2_decode_timeline.txt).
//  3. the new key comes from LDS; the 8 waves' (m, l, acc) states are merged through LDS.
// Probabilities are rounded to bf16 for the P.V product and kept in fp32 for the row sum, as in the prefill kernel.
// WPE = waves per SIMD the register allocation leaves room for: 2 (151 VGPRs: this kernel alone on its CU) or 4 (128 VGPRs, 21 dwords spilled: a second 8-wave
// block - another decode loop's attention, or its q|k|v / down_proj - fits beside it; option decode_attn_occ2, A/B in profiles/round6_attn_occupancy.txt)
template <typename T, int WPE = 2>
__global__ __launch_bounds__(512, WPE) void decode_attn_kernel(DecodeAttnArgs a) {
    typedef typename ET<T>::v8 V8;
    constexpr int HD = 128, HALF = 64, GMAX = 4, NW = 8;
    __shared__ __attribute__((aligned(16))) float s_acc[NW][GMAX][HD];
    __shared__ float s_m[NW][GMAX], s_l[NW][GMAX];
    __shared__ __attribute__((aligned(16))) T s_q[GMAX + 2][HD];   // q heads (unscaled), then k, v of the new token (all T values)
    __shared__ int s_ok[OUTL_CAP];                                 // int8 mode: the row's outlier pairs (int8_util.h)
    __shared__ float s_ox[OUTL_CAP];
    const int G = a.Hq / a.Hkv;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;                        // S: head column r, keys 4g..4g+3;  K rows: key r, dim chunk g;  V: piece r of keys 4g..4g+3
    const int b = blockIdx.x, kvh = blockIdx.y;
    if ((int)blockIdx.y >= a.Hkv) {                 // idle-CU prefetch blocks (experiment, option decode_prefetch): weights of later kernels -> L2 / Infinity Cache
        const int blk = (blockIdx.y - a.Hkv) * gridDim.x + blockIdx.x, nblk = (gridDim.y - a.Hkv) * gridDim.x;
        prefetch_share(a.pf[0], blk, nblk);
        prefetch_share(a.pf[1], blk, nblk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    T* Kc = (T*)a.Kc + ((long)b * a.Hkv + kvh) * a.ctx_max * HD;
    T* Vc = (T*)a.Vc + ((long)b * a.Hkv + kvh) * a.ctx_max * HD;
    const int cm1 = a.ctx_max - 1;

    KT(a, 0);
    V8 kf[4], vv[4], kfn[4], vvn[4];
    auto load = [&](int k0, V8 (&kd)[4], V8 (&vd)[4]) {
        const int kr = min(k0 + r, cm1);
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) kd[hs] = *(const V8*)(Kc + (long)kr * HD + hs * 32 + g * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) vd[j] = *(const V8*)(Vc + (long)min(k0 + 4 * g + j, cm1) * HD + r * 8);
    };
    load(wid * 16, kf, vv);

    int n;                                   // keys visible, the new token included at position n-1
    if (a.P && !a.dq.sca) {
        // 16-bit kinds (round 5): four consecutive dims per thread.  Round 4's form below (kept for the int8 kind, whose dequantisation owns single
        // columns) had 384 threads fetch two scalars per slab - and eight slabs whatever ksplit was, the spare ones as dummies: 96 wave-wide 4-byte
        // loads per block behind the 64 of the first K/V slice, on a CU that issues one load instruction per ~15 ns.  Now (G + 2) * 16 threads fetch
        // 2 x ksplit 16-byte pieces: 16 instructions at ksplit = 4.  Same sums in the same order, same RoPE roundings: same bits.
        typedef typename ET<T>::v4 V4;
        const int N = (a.Hq + 2 * a.Hkv) * HD;
        const int vi = tid >> 4, i4 = (tid & 15) * 4;        // vector (q heads.., k, v), first of this thread's dims in the first half
        const bool act = tid < (G + 2) * 16;
        const int col = act ? (vi < G ? (kvh * G + vi) : vi == G ? (a.Hq + kvh) : (a.Hq + a.Hkv + kvh)) * HD + i4 : 0;
        f32x4 v1[8], v2[8];
        if (act) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                if (ks < a.ksplit) {
                    const float* p = a.P + ((long)ks * a.mpad + b) * N + col;
                    v1[ks] = *(const f32x4*)p; v2[ks] = *(const f32x4*)(p + HALF);
                }
        }
        n = min(max(a.kv_len[b], 1), a.ctx_max);      // (clamped: the cache / RoPE table rows of this block end at ctx_max)
        if (act) {
            f32x4 x1 = {0.f, 0.f, 0.f, 0.f}, x2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                if (ks < a.ksplit) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x1[e] += v1[ks][e]; x2[e] += v2[ks][e]; }
                }
            const int pos = n - 1;
            f32x4 c = {1.f, 1.f, 1.f, 1.f}, sn = {0.f, 0.f, 0.f, 0.f};
            if (vi <= G) { c = *(const f32x4*)(a.cs + (long)pos * HD + i4); sn = *(const f32x4*)(a.cs + (long)pos * HD + HALF + i4); }
            V4 b1, b2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y1 = rT<T>(x1[e]), y2 = rT<T>(x2[e]);
                float o1 = y1, o2 = y2;
                if (vi <= G) {
                    o1 = rT<T>(rT<T>(y1 * c[e]) + rT<T>(-y2 * sn[e]));
                    o2 = rT<T>(rT<T>(y2 * c[e]) + rT<T>(y1 * sn[e]));
                }
                b1[e] = (T)o1; b2[e] = (T)o2;
            }
            const int row = vi < G ? vi : (vi == G ? GMAX : GMAX + 1);
            *(V4*)&s_q[row][i4] = b1; *(V4*)&s_q[row][HALF + i4] = b2;
            if (vi == G) { *(V4*)(Kc + (long)pos * HD + i4) = b1; *(V4*)(Kc + (long)pos * HD + HALF + i4) = b2; }
            if (vi == G + 1) { *(V4*)(Vc + (long)pos * HD + i4) = b1; *(V4*)(Vc + (long)pos * HD + HALF + i4) = b2; }
        }
        __syncthreads();
    } else if (a.P) {
        // QKV slab sum of this (segment, kv head): issued before kv_len is needed, so the two latencies overlap
        const int N = (a.Hq + 2 * a.Hkv) * HD;
        const int w = tid, vi = w / HALF, i = w % HALF;      // vector (q heads.., k, v), index in the first half; (G + 2) * 64 <= 384 threads
        const bool act = w < (G + 2) * HALF;
        float x1 = 0.f, x2 = 0.f;
        const int col = act ? (vi < G ? (kvh * G + vi) : vi == G ? (a.Hq + kvh) : (a.Hq + a.Hkv + kvh)) * HD + i : 0;
        if (a.dq.sca) {
            // int8 mode: int32 slabs of the quantised q/k/v projections -> fp16 module outputs (LLM.int8 dequant + outliers); the
            // row's outlier pairs are requested with the slabs and shared through LDS
            const OutlStage os = outl_issue(a.dq, b);
            const Slab1x2 sl = slab1x2_load(a.dq, a.P, a.ksplit, a.mpad, b, col, HALF, N);
            outl_commit(a.dq, b, os, s_ok, s_ox);
            if (act) slab1x2_finish(a.dq, a.P, a.ksplit, a.mpad, b, col, HALF, N, sl, os, s_ok, s_ox, x1, x2);
        }
        if (act) {
            if (a.dq.sca) {
            } else {
                // all slab loads in flight at once (a rolled loop would serialise one L2 round trip per slab); fixed summation order
                float v1[8], v2[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const float* p = a.P + ((long)(ks < a.ksplit ? ks : 0) * a.mpad + b) * N + col;
                    v1[ks] = p[0]; v2[ks] = p[HALF];
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    if (ks < a.ksplit) { x1 += v1[ks]; x2 += v2[ks]; }
            }
        }
        n = min(max(a.kv_len[b], 1), a.ctx_max);      // (clamped: the cache / RoPE table rows of this block end at ctx_max)
        if (act) {
            const int pos = n - 1;
            x1 = rT<T>(x1); x2 = rT<T>(x2);
            float o1 = x1, o2 = x2;
            if (vi <= G) {
                const float c = a.cs[(long)pos * HD + i], sn = a.cs[(long)pos * HD + HALF + i];
                o1 = rT<T>(rT<T>(x1 * c) + rT<T>(-x2 * sn));
                o2 = rT<T>(rT<T>(x2 * c) + rT<T>(x1 * sn));
            }
            const int row = vi < G ? vi : (vi == G ? GMAX : GMAX + 1);
            const T b1 = (T)o1, b2 = (T)o2;
            s_q[row][i] = b1; s_q[row][HALF + i] = b2;
            if (vi == G) { Kc[(long)pos * HD + i] = b1; Kc[(long)pos * HD + HALF + i] = b2; }
            if (vi == G + 1) { Vc[(long)pos * HD + i] = b1; Vc[(long)pos * HD + HALF + i] = b2; }
        }
        __syncthreads();
    } else {
        n = min(max(a.kv_len[b], 1), a.ctx_max);
    }
    KT(a, 1);
    // q as the MFMA B operand: column r = head (zero beyond the group), k = head-dim
    V8 qf[4];
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) {
        const V8 t = a.P ? *(const V8*)&s_q[r < G ? r : 0][hs * 32 + g * 8]
                         : *(const V8*)((const T*)a.Q + (long)b * a.Hq * HD + (kvh * G + (r < G ? r : 0)) * HD + hs * 32 + g * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[hs][j] = r < G ? t[j] : (T)0.f;
    }
    float m_r = -1e30f, lsum = 0.f;          // running max / partial row sum of head r over this lane's keys
    f32x4 acc[8];                            // O tiles: acc[t][j] = head 4g + j (real heads: g == 0), head-dim column 8r + t
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto step = [&](int k0, const V8 (&kd)[4], const V8 (&vd)[4], int limit) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) st = ET<T>::mfma(kd[hs], qf[hs], st);
        float sc[4], mx = -1e30f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = (k0 + g * 4 + j) < limit ? st[j] * a.scale : -1e30f; mx = fmaxf(mx, sc[j]); }
        mx = rows_max(mx);                                   // over the four key quarters: the slice maximum of head r, in every lane of column r
        const float mn = fmaxf(m_r, mx);
        const float alpha = __expf(m_r - mn);
        m_r = mn;
        float ps = 0.f, pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float pv = ((k0 + g * 4 + j) < limit && r < G) ? __expf(sc[j] - mn) : 0.f; ps += pv; pr[j] = pv; }
        lsum = lsum * alpha + ps;
        u32x2_t pa;                                          // A fragment: P[head r][keys 4g..4g+3], rounded to T
        {
            typename ET<T>::v4 p4;
#pragma unroll
            for (int j = 0; j < 4; ++j) p4[j] = (T)pr[j];
            pa = __builtin_bit_cast(u32x2_t, p4);
        }
        // rescale the accumulators: row j of a tile is head j, whose factor sits in lane j (column j, first key quarter)
        const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), 0));
        const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), 1));
        const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), 2));
        const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), 3));
        const unsigned* w0 = (const unsigned*)&vd[0]; const unsigned* w1 = (const unsigned*)&vd[1];
        const unsigned* w2 = (const unsigned*)&vd[2]; const unsigned* w3 = (const unsigned*)&vd[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // tiles 2i (low halves of dword i of the four pieces) and 2i + 1 (high halves): B fragment = element t of keys 4g..4g+3
            u32x2_t blo, bhi;
            blo[0] = __builtin_amdgcn_perm(w1[i], w0[i], 0x05040100u); blo[1] = __builtin_amdgcn_perm(w3[i], w2[i], 0x05040100u);
            bhi[0] = __builtin_amdgcn_perm(w1[i], w0[i], 0x07060302u); bhi[1] = __builtin_amdgcn_perm(w3[i], w2[i], 0x07060302u);
            f32x4 c0 = acc[2 * i], c1 = acc[2 * i + 1];
            c0[0] *= a0; c0[1] *= a1; c0[2] *= a2; c0[3] *= a3;
            c1[0] *= a0; c1[1] *= a1; c1[2] *= a2; c1[3] *= a3;
            acc[2 * i] = PV16<T>::mfma(pa, blo, c0);
            acc[2 * i + 1] = PV16<T>::mfma(pa, bhi, c1);
        }
    };
    const int nc = a.P ? n - 1 : n;          // keys read from the cache
    for (int k0 = wid * 16; k0 < nc; k0 += NW * 16) {
        const bool more = k0 + NW * 16 < nc;
        if (more) load(k0 + NW * 16, kfn, vvn);
        step(k0, kf, vv, nc);
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { kf[u] = kfn[u]; vv[u] = vvn[u]; }
        }
    }
    if (a.P && wid == NW - 1) {              // the token being decoded: its k / v are still in LDS (slice key 0 only); the last wave has
                                             // the fewest cached slices (slices go round-robin from wave 0), so it takes the extra step
#pragma unroll
        for (int hs = 0; hs < 4; ++hs) kf[hs] = *(const V8*)&s_q[GMAX][hs * 32 + g * 8];
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = *(const V8*)&s_q[GMAX + 1][r * 8];
        step(0, kf, vv, 1);
    }
    KT(a, 2);
    // merge: per wave the row sums over the 4 key quarters (the outputs are already summed over the keys by the MFMA); then the 8 waves
    lsum = rows_sum(lsum);
    if (g == 0 && r < 4) { s_l[wid][r] = lsum; s_m[wid][r] = m_r; }
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4*)&s_acc[wid][j][8 * r] = (f32x4){acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
            *(f32x4*)&s_acc[wid][j][8 * r + 4] = (f32x4){acc[4][j], acc[5][j], acc[6][j], acc[7][j]};
        }
    }
    __syncthreads();
    KT(a, 3);
    float amx = 0.f;                         // int8 mode: absmax of this thread's outputs without the elements >= 6.0
    int nbig = 0;                            // ... and how many are >= 6.0
    for (int idx = tid; idx < G * HD; idx += 512) {
        const int h = idx / HD, e = idx % HD;
        float M = -1e30f;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][h]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const float f = __expf(s_m[w][h] - M); num += f * s_acc[w][h][e]; den += f * s_l[w][h]; }
        const T ov = (T)(num / den);
        ((T*)a.O)[(long)b * a.Hq * HD + (kvh * G + h) * HD + e] = ov;
        const float av = fabsf((float)ov);
        amx = fmaxf(amx, av < LLM_INT8_THRESHOLD ? av : 0.f);
        nbig += !(av < LLM_INT8_THRESHOLD);
    }
    if (a.amax_out) {
        // this block's share of the row absmax for o_proj's on-the-fly quantisation (SkinnyArgs.x_amax): one plain store per block into the
        // row's 4 partials (an atomicMax per wave onto one word per row cost 8 us per launch: 13.1 -> 21.4)
        amx = wave_max(amx);
        const int wbig = __popcll(__ballot(nbig > 0)) > 0 ? (int)wave_sum((float)nbig) : 0;
        __syncthreads();                     // s_l / s_m are free: every thread has left the merge loop
        if (lane == 0) { s_l[wid][0] = amx; s_m[wid][0] = (float)wbig; }
        __syncthreads();
        if (tid == 0) {
            float m = s_l[0][0], nb = s_m[0][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) { m = fmaxf(m, s_l[w][0]); nb += s_m[w][0]; }
            a.amax_out[b * 4 + kvh] = m;
            if (a.big_out) a.big_out[b * 4 + kvh] = (int)nb;
        }
    }
    KT(a, 4);
}

void launch_decode_attn(const DecodeAttnArgs& a, int B, hipStream_t s) {
#ifdef SONIC_AB
    if (g_opts.decode_attn_v1) { DT_SWITCH(a.dt, T, hipLaunchKernelGGL(decode_attn_v1_kernel<T>, dim3(B, a.Hkv), dim3(512), 0, s, a)); return; }
#endif
    if (g_opts.decode_attn_occ2) { DT_SWITCH(a.dt, T, hipLaunchKernelGGL((decode_attn_kernel<T, 4>), dim3(B, a.Hkv), dim3(512), 0, s, a)); return; }
    DT_SWITCH(a.dt, T, hipLaunchKernelGGL((decode_attn_kernel<T, 2>), dim3(B, a.Hkv + (a.pf_y > 0 ? a.pf_y : 0)), dim3(512), 0, s, a));
}
