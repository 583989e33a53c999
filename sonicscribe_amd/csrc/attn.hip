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


template <int HD> __device__ __forceinline__ int kswz(int row) {  // swizzle term for the K tile (rows = keys)
    if (HD == 64) return ((row >> 1) & 1) | (((row >> 3) & 3) << 1);
    return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void flash_attn_kernel(FlashArgs a) {
    constexpr int HS = HD / 32;        // hd k-steps for S^T
    constexpr int HB = HD / 16;        // hd blocks of O^T
    constexpr int KCH = HD / 8;        // 16-B chunks per K row
    constexpr int KROW = HD * 2;       // bytes per K row
    constexpr int KT_BYTES = 64 * KROW;
    constexpr int VT_BYTES = HD * 128;
    constexpr int KV_PASSES = KT_BYTES / 4096;  // 256 threads x 16 B per pass
    __shared__ __attribute__((aligned(16))) char smem[KT_BYTES + VT_BYTES];
    char* sK = smem;
    char* sV = smem + KT_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, kh = h / (a.Hq / a.Hkv);
    const int q_len = a.q_len ? a.q_len[b] : a.T;
    const int kv_len = a.kv_len ? a.kv_len[b] : a.T;
    const int q0 = blockIdx.x * 128;
    if (q0 >= q_len) return;
    const long qbase = a.q_off ? (long)a.q_off[b] * a.q_ld : (long)b * a.q_seq_stride;
    const long obase = a.q_off ? (long)a.q_off[b] * a.o_ld : (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld;
    const bf16_t* Q = a.Q + qbase + (long)h * HD;
    const bf16_t* K = a.K + (long)b * a.k_seq_stride + (long)kh * a.k_head_stride;
    const bf16_t* Vt = a.Vt + (long)b * a.vt_seq_stride + (long)kh * a.vt_head_stride;
    const int qpos_off = kv_len - q_len;  // absolute position of query 0 (causal)

    // Q fragments (B-operand): lane holds Q[query = qb*16 + fr][hd = hs*32 + fg*8 .. +7]
    bf16x8 qf[2][HS];
    int qrow[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = q0 + wid * 32 + qb * 16 + fr;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) qf[qb][hs] = *(const bf16x8*)(Q + (long)qr * a.q_ld + hs * 32 + fg * 8);
    }

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
    bf16x8 kreg[KV_PASSES], vreg[KV_PASSES];
    auto issue = [&](int kt) {
        const int key0 = kt * 64;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid;
            const int kr = idx / KCH, kc = idx % KCH;        // K tile: row = key, chunk of 8 hd
            kreg[p] = *(const bf16x8*)(K + (long)(key0 + kr) * a.k_ld + kc * 8);
            const int vr = idx >> 3, vc = idx & 7;           // V^T tile: row = hd, chunk of 8 keys
            vreg[p] = *(const bf16x8*)(Vt + (long)vr * a.vt_ld + key0 + vc * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid;
            const int kr = idx / KCH, kc = idx % KCH;
            *(bf16x8*)(sK + kr * KROW + ((kc ^ kswz<HD>(kr)) << 4)) = kreg[p];
            const int vr = idx >> 3, vc = idx & 7;
            *(bf16x8*)(sV + vr * 128 + ((vc ^ (vr & 7)) << 4)) = vreg[p];
        }
    };

    issue(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        __syncthreads();          // previous tile fully consumed
        commit();
        __syncthreads();
        if (kt + 1 < n_tiles) issue(kt + 1);
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
                    const bf16x8 kf = *(const bf16x8*)(sK + krow * KROW + ((c ^ kswz<HD>(krow)) << 4));
                    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][hs], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][hs], s1, 0, 0, 0);
                }
                st[ks][sb][0] = s0;
                st[ks][sb][1] = s1;
            }

        // ---- online softmax per query (= per lane), P^T fragments
        const bool edge = (key0 + 64 > kv_len) || (CAUSAL && (key0 + 63 > q0 + wid * 32 + qpos_off));
        bf16x8 pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -1e30f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float s = st[ks][sb][qb][j] * a.scale;
                        if (edge) {
                            const int key = key0 + ks * 32 + 8 * fg + 4 * sb + j;
                            bool ok = key < kv_len;
                            if (CAUSAL) ok = ok && (key <= qrow[qb] + qpos_off);
                            s = ok ? s : -1e30f;
                        }
                        st[ks][sb][qb][j] = s;
                        mx = fmaxf(mx, s);
                    }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrun[qb], mx);
            const float alpha = __expf(mrun[qb] - mnew);
            mrun[qb] = mnew;
            float psum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float p = __expf(st[ks][sb][qb][j] - mnew);
                        psum += p;
                        pf[qb][ks][sb * 4 + j] = f2bf(p);
                    }
            lrun[qb] = lrun[qb] * alpha + psum;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] *= alpha;
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const int vr = hb * 16 + fr, c = ks * 4 + fg;
                const bf16x8 vf = *(const bf16x8*)(sV + vr * 128 + ((c ^ (vr & 7)) << 4));
                oacc[0][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][ks], oacc[0][hb], 0, 0, 0);
                oacc[1][hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][ks], oacc[1][hb], 0, 0, 0);
            }
    }

    // ---- epilogue: O[query][h*HD + hb*16 + fg*4 + j] = O^T / l
    bf16_t* O = a.O + obase + (long)h * HD;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float l = lrun[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (qrow[qb] < q_len) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(oacc[qb][hb][j] / l);
                *(bf16x4*)(O + (long)qrow[qb] * a.o_ld + hb * 16 + fg * 4) = o;
            }
        }
    }
}

void launch_flash(const FlashArgs& a, int hd, bool causal, int B, int max_q, hipStream_t s) {
    dim3 grid((max_q + 127) / 128, a.Hq, B), block(256);
    if (hd == 64 && !causal) hipLaunchKernelGGL((flash_attn_kernel<64, false>), grid, block, 0, s, a);
    else if (hd == 64 && causal) hipLaunchKernelGGL((flash_attn_kernel<64, true>), grid, block, 0, s, a);
    else if (hd == 128 && causal) hipLaunchKernelGGL((flash_attn_kernel<128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((flash_attn_kernel<128, false>), grid, block, 0, s, a);
}

// ------------------------------------------------------------------------------------------------

// HD = 128, group size G = Hq/Hkv <= 4.  One block (8 waves) per (sequence, kv head).
//  1. prologue (fused RoPE + KV append, modeling_llama.py:121-143,261-262): sums the QKV skinny-GEMM slabs of this
//     (sequence, kv head), rounds to bf16, applies rotate-half RoPE at the token's position, writes the new K / V rows
//     into the cache and keeps q (4 heads), k, v in LDS;
//  2. single pass over the cached keys with a per-wave online softmax: a wave-iteration covers 4 x 4 keys of K and V
//     (8 KiB per wave, next iteration's rows prefetched into registers before the current one is consumed), lane
//     (sub = lane>>4, ch = lane&15) owns key `sub` of each 4-key group and head-dim chunk `ch`;
//  3. the new key comes from LDS; the 8 waves' (m, l, acc) states are merged through LDS.
// Probabilities are rounded to bf16 for the P.V product and kept in fp32 for the row sum, as in the prefill kernel.
__global__ __launch_bounds__(512) void decode_attn_kernel(DecodeAttnArgs a) {
    constexpr int HD = 128, HALF = 64, GMAX = 4, U = 4, NW = 8;
    __shared__ float s_acc[NW][GMAX][HD];
    __shared__ float s_m[NW][GMAX], s_l[NW][GMAX];
    __shared__ float s_q[GMAX + 2][HD];     // q heads (scaled), then k, v of the new token
    const int G = a.Hq / a.Hkv;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x, kvh = blockIdx.y;
    const int n = a.kv_len[b];              // keys visible, the new token included at position n-1
    const int sub = lane >> 4, ch = lane & 15;
    bf16_t* Kc = a.Kc + ((long)b * a.Hkv + kvh) * a.ctx_max * HD;
    bf16_t* Vc = a.Vc + ((long)b * a.Hkv + kvh) * a.ctx_max * HD;

    if (a.P) {
        const int N = (a.Hq + 2 * a.Hkv) * HD, pos = n - 1;
        for (int w = tid; w < (G + 2) * HALF; w += 512) {
            const int vi = w / HALF, i = w % HALF;          // vector (q heads.., k, v), index in the first half
            const int col = (vi < G ? (kvh * G + vi) : vi == G ? (a.Hq + kvh) : (a.Hq + a.Hkv + kvh)) * HD + i;
            float x1 = 0.f, x2 = 0.f;
            for (int ks = 0; ks < a.ksplit; ++ks) {
                const float* p = a.P + ((long)ks * a.mpad + b) * N + col;
                x1 += p[0]; x2 += p[HALF];
            }
            x1 = rbf(x1); x2 = rbf(x2);
            float o1 = x1, o2 = x2;
            if (vi <= G) {
                const float c = a.cs[(long)pos * HD + i], sn = a.cs[(long)pos * HD + HALF + i];
                o1 = rbf(rbf(x1 * c) + rbf(-x2 * sn));
                o2 = rbf(rbf(x2 * c) + rbf(x1 * sn));
            }
            const float sc = vi < G ? a.scale : 1.0f;
            s_q[vi < G ? vi : (vi == G ? GMAX : GMAX + 1)][i] = o1 * sc;
            s_q[vi < G ? vi : (vi == G ? GMAX : GMAX + 1)][HALF + i] = o2 * sc;
            if (vi == G) { Kc[(long)pos * HD + i] = f2bf(o1); Kc[(long)pos * HD + HALF + i] = f2bf(o2); }
            if (vi == G + 1) { Vc[(long)pos * HD + i] = f2bf(o1); Vc[(long)pos * HD + HALF + i] = f2bf(o2); }
        }
        __syncthreads();
    }
    float q[GMAX][8];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (a.P) {
#pragma unroll
            for (int i = 0; i < 8; ++i) q[g][i] = s_q[g < G ? g : 0][ch * 8 + i];
        } else {
            const int hq = kvh * G + (g < G ? g : 0);
            const bf16x8 v = *(const bf16x8*)(a.Q + (long)b * a.Hq * HD + hq * HD + ch * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) q[g][i] = bf2f(v[i]) * a.scale;
        }
    }
    float m[GMAX], l[GMAX], acc[GMAX][8];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        m[g] = -1e30f; l[g] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[g][i] = 0.f;
    }
    const int nc = a.P ? n - 1 : n;          // keys read from the cache
    const bf16_t* Kr = Kc + ch * 8;
    const bf16_t* Vr = Vc + ch * 8;
    bf16x8 kv[U], vv[U], kn[U], vn[U];
    auto load = [&](int k0, bf16x8 (&kd)[U], bf16x8 (&vd)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int key = k0 + u * 4 + sub; key = key < nc ? key : (nc > 0 ? nc - 1 : 0);
            kd[u] = *(const bf16x8*)(Kr + (long)key * HD);
            vd[u] = *(const bf16x8*)(Vr + (long)key * HD);
        }
    };
    auto consume = [&](int k0, const bf16x8 (&kd)[U], const bf16x8 (&vd)[U], int limit) {
        float sc[U][GMAX];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = (k0 + u * 4 + sub) < limit;
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) d += q[g][i] * bf2f(kd[u][i]);
                d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
                sc[u][g] = ok[u] ? d : -1e30f;
            }
        }
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            float mx = fmaxf(fmaxf(sc[0][g], sc[1][g]), fmaxf(sc[2][g], sc[3][g]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[g], mx);
            const float alpha = __expf(m[g] - mn);
            m[g] = mn;
            float ps = 0.f, pr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const float p = ok[u] ? __expf(sc[u][g] - mn) : 0.f; ps += p; pr[u] = rbf(p); }
            l[g] = l[g] * alpha + ps;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = acc[g][i] * alpha;
#pragma unroll
                for (int u = 0; u < U; ++u) t += pr[u] * bf2f(vd[u][i]);
                acc[g][i] = t;
            }
        }
    };
    const int stride = NW * 4 * U;
    int k0 = wid * (4 * U);
    if (k0 < nc) load(k0, kv, vv);
    for (; k0 < nc; k0 += stride) {
        const bool more = k0 + stride < nc;
        if (more) load(k0 + stride, kn, vn);
        consume(k0, kv, vv, nc);
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) { kv[u] = kn[u]; vv[u] = vn[u]; }
        }
    }
    if (a.P && wid == 0) {                   // the token being decoded: its k / v are still in LDS
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) { kv[u][i] = f2bf(s_q[GMAX][ch * 8 + i]); vv[u][i] = f2bf(s_q[GMAX + 1][ch * 8 + i]); }
        consume(0, kv, vv, 1);                // only (u = 0, sub = 0) is in range
    }
    // merge the 4 key sub-groups of the wave, then the 8 waves
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        float lv = l[g];
        lv += __shfl_xor(lv, 16, 64); lv += __shfl_xor(lv, 32, 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = acc[g][i];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (sub == 0) s_acc[wid][g][ch * 8 + i] = v;
        }
        if (lane == 0) { s_m[wid][g] = m[g]; s_l[wid][g] = lv; }
    }
    __syncthreads();
    for (int idx = tid; idx < G * HD; idx += 512) {
        const int g = idx / HD, e = idx % HD;
        float M = -1e30f;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, s_m[w][g]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const float f = __expf(s_m[w][g] - M); num += f * s_acc[w][g][e]; den += f * s_l[w][g]; }
        a.O[(long)b * a.Hq * HD + (kvh * G + g) * HD + e] = f2bf(num / den);
    }
}

void launch_decode_attn(const DecodeAttnArgs& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(decode_attn_kernel, dim3(B, a.Hkv), dim3(512), 0, s, a);
}
