// What bounds the encoder's flash attention (head dim 64, T = 1500, 20 heads, 32 sequences: 513 us per layer, 0.29 of the bf16 MFMA peak)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sonicscribe_amd/csrc tools/flash_bound.hip -o tools/flash_bound && tools/flash_bound
// Times the product kernel and copies of its loop with parts removed (the results of those are wrong on purpose) and a software-pipelined
// form (S of key tile kt+1 issued before the softmax of tile kt, three LDS tile buffers, one barrier per tile; same arithmetic, same bits).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../sonicscribe_amd/csrc/attn.hip"
thread_local LaunchOpts g_opts;
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// MODE 0: the product loop;  1: no softmax (P = bf16(S));  2: no MFMA (LDS reads kept);  3: neither (loads, LDS traffic, barriers only)
template <int MODE>
__global__ __launch_bounds__(256) void flash_parts_kernel(FlashArgs a) {
    typedef bf16_t T;
    constexpr int HD = 64;
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    constexpr int HS = HD / 32, HB = HD / 16, KCH = HD / 8, KROW = HD * 2, KT_BYTES = 64 * KROW, VT_BYTES = HD * 128, KV_PASSES = KT_BYTES / 4096;
    __shared__ __attribute__((aligned(16))) char smem[KT_BYTES + VT_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q_len = a.T, kv_len = a.T, q0 = blockIdx.x * 128;
    if (q0 >= q_len) return;
    const T* Q = (const T*)a.Q + (long)b * a.q_seq_stride + (long)h * HD;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)h * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)h * a.vt_head_stride;
    V8 qf[2][HS];
    int qrow[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = q0 + wid * 32 + qb * 16 + fr;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) qf[qb][hs] = *(const V8*)(Q + (long)qr * a.q_ld + hs * 32 + fg * 8);
    }
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
    const int n_tiles = (kv_len + 63) / 64;
    V8 kreg[KV_PASSES], vreg[KV_PASSES];
    auto issue = [&](int kt) {
        const int key0 = kt * 64;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid, kr = idx / KCH, kc = idx % KCH, vr = idx >> 3, vc = idx & 7;
            kreg[p] = *(const V8*)(K + (long)(key0 + kr) * a.k_ld + kc * 8);
            vreg[p] = *(const V8*)(Vt + (long)vr * a.vt_ld + key0 + vc * 8);
        }
    };
    auto commit = [&]() {
        char* sK = smem; char* sV = sK + KT_BYTES;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid, kr = idx / KCH, kc = idx % KCH, vr = idx >> 3, vc = idx & 7;
            *(V8*)(sK + kr * KROW + ((kc ^ kswz<HD>(kr)) << 4)) = kreg[p];
            *(V8*)(sV + vr * 128 + ((vc ^ (vr & 7)) << 4)) = vreg[p];
        }
    };
    issue(0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        __syncthreads();
        commit();
        __syncthreads();
        if (kt + 1 < n_tiles) issue(kt + 1);
        const char* sK = smem; const char* sV = sK + KT_BYTES;
        const int key0 = kt * 64;
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
                    V8 kf = *(const V8*)(sK + krow * KROW + ((c ^ kswz<HD>(krow)) << 4));
                    if (MODE & 2) { asm volatile("" :: "v"(kf)); s0[0] += 1.0f; s1[1] += 1.0f; }
                    else { s0 = ET<T>::mfma(kf, qf[0][hs], s0); s1 = ET<T>::mfma(kf, qf[1][hs], s1); }
                }
                st[ks][sb][0] = s0; st[ks][sb][1] = s1;
            }
        const bool edge = key0 + 64 > kv_len;
        V8 pf[2][2];
        const float cexp = a.scale * 1.44269504088896341f;
        float alph[2] = {1.f, 1.f};
        if (MODE & 1) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) pf[qb][ks][sb * 4 + j] = (T)st[ks][sb][qb][j];
        } else {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                if (edge) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int key = key0 + ks * 32 + 8 * fg + 4 * sb + j;
                                st[ks][sb][qb][j] = key < kv_len ? st[ks][sb][qb][j] : -1e30f;
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
            if (__builtin_amdgcn_ballot_w64(alph[0] != 1.0f || alph[1] != 1.0f) != 0) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] *= alph[qb];
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const int vr = hb * 16 + fr, c = ks * 4 + fg;
                V8 vf = *(const V8*)(sV + vr * 128 + ((c ^ (vr & 7)) << 4));
                if (MODE & 2) { asm volatile("" :: "v"(vf), "v"(pf[0][ks]), "v"(pf[1][ks])); oacc[0][hb][0] += 1.0f; oacc[1][hb][0] += 1.0f; }
                else { oacc[0][hb] = ET<T>::mfma(vf, pf[0][ks], oacc[0][hb]); oacc[1][hb] = ET<T>::mfma(vf, pf[1][ks], oacc[1][hb]); }
            }
    }
    T* O = (T*)a.O + (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld + (long)h * HD;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = (MODE & 1) ? 1.0f : rows_sum(lrun[qb]);
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

// Software-pipelined form.  Iteration kt:  barrier | commit tile kt+2 (registers -> LDS buffer (kt+2) % 3), request tile kt+3 |
// S(kt+1) = K(kt+1) . Q^T  [MFMA]  beside  softmax(kt) [VALU]  |  O += V(kt) . P(kt) [MFMA].  The wave's own MFMAs and VALU work are
// independent inside an iteration, so the matrix pipe runs under the softmax instead of beside an idle VALU.
//   SCHED 0: instruction order left to the compiler;  1: sched_group_barrier pattern (one MFMA, then a slice of the softmax)
template <int SCHED>
__global__ __launch_bounds__(256, 2) void flash_pipe_kernel(FlashArgs a) {
    typedef bf16_t T;
    constexpr int HD = 64;
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    constexpr int HS = HD / 32, HB = HD / 16, KCH = HD / 8, KROW = HD * 2, KT_BYTES = 64 * KROW, VT_BYTES = HD * 128, KV_PASSES = KT_BYTES / 4096;
    constexpr int BUF = KT_BYTES + VT_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[3 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q_len = a.T, kv_len = a.T, q0 = blockIdx.x * 128;
    if (q0 >= q_len) return;
    const T* Q = (const T*)a.Q + (long)b * a.q_seq_stride + (long)h * HD;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)h * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)h * a.vt_head_stride;
    V8 qf[2][HS];
    int qrow[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = q0 + wid * 32 + qb * 16 + fr;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) qf[qb][hs] = *(const V8*)(Q + (long)qr * a.q_ld + hs * 32 + fg * 8);
    }
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
    const int n_tiles = (kv_len + 63) / 64;
    V8 kreg[KV_PASSES], vreg[KV_PASSES];
    auto issue = [&](int kt) {
        const int key0 = kt * 64;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid, kr = idx / KCH, kc = idx % KCH, vr = idx >> 3, vc = idx & 7;
            kreg[p] = *(const V8*)(K + (long)(key0 + kr) * a.k_ld + kc * 8);
            vreg[p] = *(const V8*)(Vt + (long)vr * a.vt_ld + key0 + vc * 8);
        }
    };
    auto commit = [&](int buf) {
        char* sK = smem + buf * BUF; char* sV = sK + KT_BYTES;
#pragma unroll
        for (int p = 0; p < KV_PASSES; ++p) {
            const int idx = p * 256 + tid, kr = idx / KCH, kc = idx % KCH, vr = idx >> 3, vc = idx & 7;
            *(V8*)(sK + kr * KROW + ((kc ^ kswz<HD>(kr)) << 4)) = kreg[p];
            *(V8*)(sV + vr * 128 + ((vc ^ (vr & 7)) << 4)) = vreg[p];
        }
    };
    auto scores = [&](int buf, f32x4 (&st)[2][2][2]) {
        const char* sK = smem + buf * BUF;
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
                st[ks][sb][0] = s0; st[ks][sb][1] = s1;
            }
    };
    issue(0); commit(0);
    if (n_tiles > 1) { issue(1); commit(1); }
    if (n_tiles > 2) issue(2);
    __syncthreads();
    f32x4 st[2][2][2], sn[2][2][2];
    scores(0, st);
    const float cexp = a.scale * 1.44269504088896341f;
    for (int kt = 0; kt < n_tiles; ++kt) {
        __syncthreads();                 // tile kt+1 (committed in iteration kt-1) is visible; every wave is done with tile kt-1
        if (kt + 2 < n_tiles) { commit((kt + 2) % 3); if (kt + 3 < n_tiles) issue(kt + 3); }
        const int key0 = kt * 64;
        const bool edge = key0 + 64 > kv_len;
        V8 pf[2][2];
        float alph[2];
        if (kt + 1 < n_tiles) scores((kt + 1) % 3, sn);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (edge) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int key = key0 + ks * 32 + 8 * fg + 4 * sb + j;
                            st[ks][sb][qb][j] = key < kv_len ? st[ks][sb][qb][j] : -1e30f;
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
        if (SCHED == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one K fragment read every other MFMA (8 reads feed 16 MFMAs)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);     // a slice of the softmax
            }
        }
        if (__builtin_amdgcn_ballot_w64(alph[0] != 1.0f || alph[1] != 1.0f) != 0) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] *= alph[qb];
        }
        const char* sV = smem + (kt % 3) * BUF + KT_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const int vr = hb * 16 + fr, c = ks * 4 + fg;
                const V8 vf = *(const V8*)(sV + vr * 128 + ((c ^ (vr & 7)) << 4));
                oacc[0][hb] = ET<T>::mfma(vf, pf[0][ks], oacc[0][hb]);
                oacc[1][hb] = ET<T>::mfma(vf, pf[1][ks], oacc[1][hb]);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) { st[ks][sb][0] = sn[ks][sb][0]; st[ks][sb][1] = sn[ks][sb][1]; }
    }
    T* O = (T*)a.O + (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld + (long)h * HD;
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


// LDS-DMA ring: K / V^T tiles go HBM -> LDS without a register stop (global_load_lds, 16 B per lane; the XOR swizzle is applied on the
// global side: a lane fetches the chunk that belongs in the LDS slot it is going to write), NST stages, tile kt + NST - 1 requested while
// tile kt is consumed, counted vmcnt waits and ONE raw barrier per tile - the GEMM kernels' operand ring.  Same arithmetic, same bits.
__device__ long long* g_trace;      // [blocks][4 waves][5]: cycles in barrier wait | S MFMAs | softmax | P.V MFMAs | tiles
__device__ __forceinline__ long long now_after(float dep) {       // core-clock time once `dep` (an MFMA result) can be read
    float t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(dep));
    asm volatile("" :: "v"(t));
    return __builtin_amdgcn_s_memtime();
}
template <int NST, int QB, bool TRACE = false>
__global__ __launch_bounds__(256, 2) void flash_ring_kernel(FlashArgs a) {
    long long tr[4] = {0, 0, 0, 0}, t_prev = 0;
    typedef bf16_t T;
    constexpr int HD = 64;
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    constexpr int HS = HD / 32, HB = HD / 16, KROW = HD * 2, KT_BYTES = 64 * KROW, VT_BYTES = HD * 128, BUF = KT_BYTES + VT_BYTES, PF = NST - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q_len = a.T, kv_len = a.T, q0 = blockIdx.x * (QB * 64);
    if (q0 >= q_len) return;
    const T* Q = (const T*)a.Q + (long)b * a.q_seq_stride + (long)h * HD;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)h * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)h * a.vt_head_stride;
    const int n_tiles = (kv_len + 63) / 64;
    // this lane's two K rows / two V^T rows of a tile and the (swizzled) chunk it fetches from each
    const T* ksrc[2]; const T* vsrc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (wid * 2 + u) * 8 + (lane >> 3), pc = lane & 7;
        ksrc[u] = K + (long)r * a.k_ld + ((pc ^ kswz<HD>(r)) << 3);
        vsrc[u] = Vt + (long)r * a.vt_ld + ((pc ^ (r & 7)) << 3);
    }
    auto load_tile = [&](int kt, int stage) {
        const int key0 = (kt < n_tiles ? kt : n_tiles - 1) * 64;
        char* sK = smem + stage * BUF; char* sV = sK + KT_BYTES;
#pragma unroll
        for (int u = 0; u < 2; ++u) glds16(ksrc[u] + (long)key0 * a.k_ld, sK + (wid * 2 + u) * 1024);
#pragma unroll
        for (int u = 0; u < 2; ++u) glds16(vsrc[u] + key0, sV + (wid * 2 + u) * 1024);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load_tile(p, p);
    V8 qf[QB][HS];
    int qrow[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = q0 + wid * (QB * 16) + qb * 16 + fr;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) qf[qb][hs] = *(const V8*)(Q + (long)qr * a.q_ld + hs * 32 + fg * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // Q fragments (and, once, the first tiles) have landed
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) asm volatile("" : "+v"(qf[qb][hs]));
    f32x4 oacc[QB][HB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[QB], lrun[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { mrun[qb] = -1e30f; lrun[qb] = 0.f; }
    const float cexp = a.scale * 1.44269504088896341f;
    for (int kt = 0; kt < n_tiles; ++kt) {
        if (TRACE && kt == 0) t_prev = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * 4) : "memory");     // this wave's pieces of tile kt are in LDS
        __builtin_amdgcn_s_barrier();                                             // ... everybody's; and tile kt - 1 is consumed
        load_tile(kt + PF, (kt + PF) % NST);
        long long t0 = 0, t1 = 0, t2 = 0;
        if (TRACE) { t0 = __builtin_amdgcn_s_memtime(); tr[0] += t0 - t_prev; }
        const char* sK = smem + (kt % NST) * BUF; const char* sV = sK + KT_BYTES;
        const int key0 = kt * 64;
        f32x4 st[2][2][QB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int krow = ks * 32 + 8 * (fr >> 2) + 4 * sb + (fr & 3);
                f32x4 sq[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) sq[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hs = 0; hs < HS; ++hs) {
                    const int c = hs * 4 + fg;
                    const V8 kf = *(const V8*)(sK + krow * KROW + ((c ^ kswz<HD>(krow)) << 4));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) sq[qb] = ET<T>::mfma(kf, qf[qb][hs], sq[qb]);
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) st[ks][sb][qb] = sq[qb];
            }
        if (TRACE) { t1 = now_after(st[1][1][QB - 1][3]); tr[1] += t1 - t0; }
        const bool edge = key0 + 64 > kv_len;
        V8 pf[QB][2];
        float alph[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (edge) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int key = key0 + ks * 32 + 8 * fg + 4 * sb + j;
                            st[ks][sb][qb][j] = key < kv_len ? st[ks][sb][qb][j] : -1e30f;
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
        if (TRACE) { t2 = now_after((float)pf[QB - 1][1][7] + lrun[QB - 1]); tr[2] += t2 - t1; }
        bool moved = false;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) moved = moved || alph[qb] != 1.0f;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] *= alph[qb];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const int vr = hb * 16 + fr, c = ks * 4 + fg;
                const V8 vf = *(const V8*)(sV + vr * 128 + ((c ^ (vr & 7)) << 4));
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) oacc[qb][hb] = ET<T>::mfma(vf, pf[qb][ks], oacc[qb][hb]);
            }
        if (TRACE) { t_prev = now_after(oacc[QB - 1][HB - 1][3]); tr[3] += t_prev - t2; }
    }
    if (TRACE && lane == 0) {
        long long* w = g_trace + ((long)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + wid) * 5;
        w[0] = tr[0]; w[1] = tr[1]; w[2] = tr[2]; w[3] = tr[3]; w[4] = n_tiles;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped tail requests: nothing may still be writing LDS when the block leaves
    T* O = (T*)a.O + (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld + (long)h * HD;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
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


// Staggered form.  A block is 8 waves = 256 queries: waves 0-3 and 4-7 are SIMD partners and run the same program HALF A KEY TILE apart, with
// one barrier per half tile: while one partner is in its MFMA-heavy half (S = K.Q^T, then the row maxima) the other is in its VALU-heavy half
// (exp2 / sums / conversion, then O += V.P).  Two waves that alternate phases on their own drift into lockstep - both wait for the matrix pipe,
// then both for the VALU (tools/mfma_valu_overlap.hip: a MFMA wave and a VALU wave on one SIMD overlap completely, two mixed waves hardly).
// K / V tiles: LDS-DMA ring of NST stages (prefetch distance NST - 2: a tile stays until the late partner is done with it).  Same arithmetic.
template <int NST, int MINB>
__global__ __launch_bounds__(512, MINB) void flash_stag_kernel(FlashArgs a) {
    typedef bf16_t T;
    constexpr int HD = 64, QB = 2;
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    constexpr int HS = HD / 32, HB = HD / 16, KROW = HD * 2, KT_BYTES = 64 * KROW, VT_BYTES = HD * 128, BUF = KT_BYTES + VT_BYTES, PF = NST - 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int lag = wid >> 2;
    const int b = blockIdx.z, h = blockIdx.y, q_len = a.T, kv_len = a.T, q0 = blockIdx.x * 256;
    if (q0 >= q_len) return;
    const T* Q = (const T*)a.Q + (long)b * a.q_seq_stride + (long)h * HD;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)h * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)h * a.vt_head_stride;
    const int n_tiles = (kv_len + 63) / 64;
    // wave w fetches K rows 8w .. 8w+7 and V^T rows 8w .. 8w+7 of every tile (one 1 KiB piece each), swizzled on the global side
    const int r8 = wid * 8 + (lane >> 3), pc = lane & 7;
    const T* ksrc = K + (long)r8 * a.k_ld + ((pc ^ kswz<HD>(r8)) << 3);
    const T* vsrc = Vt + (long)r8 * a.vt_ld + ((pc ^ (r8 & 7)) << 3);
    auto load_tile = [&](int kt) {
        const int key0 = (kt < n_tiles ? kt : n_tiles - 1) * 64;
        char* sK = smem + (kt % NST) * BUF; char* sV = sK + KT_BYTES;
        glds16(ksrc + (long)key0 * a.k_ld, sK + wid * 1024);
        glds16(vsrc + key0, sV + wid * 1024);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load_tile(p);
    V8 qf[QB][HS];
    int qrow[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        qrow[qb] = q0 + wid * (QB * 16) + qb * 16 + fr;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) qf[qb][hs] = *(const V8*)(Q + (long)qr * a.q_ld + hs * 32 + fg * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int hs = 0; hs < HS; ++hs) asm volatile("" : "+v"(qf[qb][hs]));
    f32x4 oacc[QB][HB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[QB], lrun[QB], moff[QB], alph[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { mrun[qb] = -1e30f; lrun[qb] = 0.f; moff[qb] = 0.f; alph[qb] = 1.f; }
    const float cexp = a.scale * 1.44269504088896341f;
    f32x4 st[2][2][QB];
    const int n_half = 2 * n_tiles + 1;
    for (int hstep = 0; hstep < n_half; ++hstep) {
        if (!(hstep & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * 2) : "memory");      // this wave's pieces of tile hstep / 2 have landed
        __builtin_amdgcn_s_barrier();
        if (!(hstep & 1)) load_tile((hstep >> 1) + PF);
        const int hh = hstep - lag;
        if (hh < 0 || hh >= 2 * n_tiles) continue;
        const int kt = hh >> 1, key0 = kt * 64;
        const char* sK = smem + (kt % NST) * BUF; const char* sV = sK + KT_BYTES;
        if (!(hh & 1)) {
            // ---- first half: S^T = K . Q^T, masking, running maxima
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const int krow = ks * 32 + 8 * (fr >> 2) + 4 * sb + (fr & 3);
                    f32x4 sq[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) sq[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int hs = 0; hs < HS; ++hs) {
                        const int c = hs * 4 + fg;
                        const V8 kf = *(const V8*)(sK + krow * KROW + ((c ^ kswz<HD>(krow)) << 4));
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) sq[qb] = ET<T>::mfma(kf, qf[qb][hs], sq[qb]);
                    }
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) st[ks][sb][qb] = sq[qb];
                }
            const bool edge = key0 + 64 > kv_len;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (edge) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int key = key0 + ks * 32 + 8 * fg + 4 * sb + j;
                                st[ks][sb][qb][j] = key < kv_len ? st[ks][sb][qb][j] : -1e30f;
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
                alph[qb] = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * cexp);
                mrun[qb] = mnew;
                moff[qb] = -mnew * cexp;
            }
        } else {
            // ---- second half: P = exp2(S c - m c), row sums, bf16 fragments; O^T (rescaled when a maximum moved) += V^T . P^T
            V8 pf[QB][2];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float psum = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[ks][sb][qb][j], cexp, moff[qb]));
                            psum += p;
                            pf[qb][ks][sb * 4 + j] = (T)p;
                        }
                lrun[qb] = lrun[qb] * alph[qb] + psum;
            }
            bool moved = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) moved = moved || alph[qb] != 1.0f;
            if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) oacc[qb][hb] *= alph[qb];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int hb = 0; hb < HB; ++hb) {
                    const int vr = hb * 16 + fr, c = ks * 4 + fg;
                    const V8 vf = *(const V8*)(sV + vr * 128 + ((c ^ (vr & 7)) << 4));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) oacc[qb][hb] = ET<T>::mfma(vf, pf[qb][ks], oacc[qb][hb]);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T* O = (T*)a.O + (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld + (long)h * HD;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
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

int main() {
    const int B = 32, H = 20, T = 1500, HD = 64, Tp = 1536, C = H * HD;
    const size_t nq = (size_t)B * T * C, nv = (size_t)B * C * Tp;
    bf16_t *q, *k, *vt, *o, *o2;
    hipMalloc(&q, nq * 2 + 65536); hipMalloc(&k, nq * 2 + 64 * C * 2 + 65536); hipMalloc(&vt, nv * 2 + 65536); hipMalloc(&o, nq * 2); hipMalloc(&o2, nq * 2);
    std::vector<unsigned short> h(nq > nv ? nq : nv);
    auto fill = [&](bf16_t* d, size_t n, unsigned seed, float amp) {
        unsigned s = seed;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; float f = ((int)(s >> 9) % 2001 - 1000) * (amp / 1000.f); unsigned u; std::memcpy(&u, &f, 4); h[i] = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
        hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    };
    fill(q, nq, 1, 2.0f); fill(k, nq, 2, 2.0f); fill(vt, nv, 3, 1.0f);
    FlashArgs f{};
    f.Q = q; f.q_ld = C; f.K = k; f.k_ld = C; f.Vt = vt; f.vt_ld = Tp; f.O = o; f.o_ld = C;
    f.q_seq_stride = (long)T * C; f.k_seq_stride = (long)T * C; f.k_head_stride = HD; f.vt_seq_stride = (long)C * Tp; f.vt_head_stride = (long)HD * Tp;
    f.T = T; f.Hq = H; f.Hkv = H; f.scale = 0.125f; f.dt = DT_BF16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((T + 127) / 128, H, B);
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-58s %7.1f us   (%s)\n", name, best * 1e3, hipGetErrorString(hipGetLastError()));
    };
    g_opts.flash_variant = 2;
    timeit("product kernel (flash_variant 2)", [&] { launch_flash(f, 64, false, B, T, 0); });
    g_opts.flash_variant = 3;
    timeit("  ... two LDS buffers, one barrier per tile (flash_variant 3)", [&] { launch_flash(f, 64, false, B, T, 0); });
    timeit("copy of the product loop", [&] { hipLaunchKernelGGL(flash_parts_kernel<0>, grid, dim3(256), 0, 0, f); });
    timeit("  without the softmax (P = bf16(S))", [&] { hipLaunchKernelGGL(flash_parts_kernel<1>, grid, dim3(256), 0, 0, f); });
    timeit("  without the MFMAs (LDS reads kept)", [&] { hipLaunchKernelGGL(flash_parts_kernel<2>, grid, dim3(256), 0, 0, f); });
    timeit("  without both (loads, LDS traffic, barriers)", [&] { hipLaunchKernelGGL(flash_parts_kernel<3>, grid, dim3(256), 0, 0, f); });
    FlashArgs f2 = f; f2.O = o2;
    timeit("software-pipelined, compiler's order", [&] { hipLaunchKernelGGL(flash_pipe_kernel<0>, grid, dim3(256), 0, 0, f2); });
    hipLaunchKernelGGL(flash_parts_kernel<0>, grid, dim3(256), 0, 0, f); hipDeviceSynchronize();
    std::vector<unsigned short> ha(nq), hb(nq);
    auto diff = [&](const char* name) {
        hipMemcpy(ha.data(), o, nq * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), o2, nq * 2, hipMemcpyDeviceToHost);
        size_t nd = 0; for (size_t i = 0; i < nq; ++i) nd += ha[i] != hb[i];
        printf("    %s vs the product loop: %zu of %zu outputs differ\n", name, nd, nq);
    };
    diff("pipelined (compiler's order)");
    hipMemset(o2, 0, nq * 2);
    timeit("software-pipelined, sched_group_barrier pattern", [&] { hipLaunchKernelGGL(flash_pipe_kernel<1>, grid, dim3(256), 0, 0, f2); });
    diff("pipelined (sched pattern)");
    hipMemset(o2, 0, nq * 2);
    hipFuncSetAttribute((const void*)flash_ring_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
    hipFuncSetAttribute((const void*)flash_ring_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
    hipFuncSetAttribute((const void*)flash_ring_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
    hipFuncSetAttribute((const void*)flash_ring_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
    timeit("LDS-DMA ring, 4 stages, one barrier per tile", [&] { hipLaunchKernelGGL((flash_ring_kernel<4, 2>), grid, dim3(256), 4 * 16384, 0, f2); });
    diff("ring (4 stages)");
    hipMemset(o2, 0, nq * 2);
    timeit("LDS-DMA ring, 3 stages", [&] { hipLaunchKernelGGL((flash_ring_kernel<3, 2>), grid, dim3(256), 3 * 16384, 0, f2); });
    diff("ring (3 stages)");
    const dim3 grid64((T + 255) / 256, H, B);
    hipMemset(o2, 0, nq * 2);
    timeit("LDS-DMA ring, 3 stages, 64 queries per wave (256 per block)", [&] { hipLaunchKernelGGL((flash_ring_kernel<3, 4>), grid64, dim3(256), 3 * 16384, 0, f2); });
    diff("ring (3 stages, 64 queries per wave)");
    hipMemset(o2, 0, nq * 2);
    timeit("LDS-DMA ring, 4 stages, 64 queries per wave", [&] { hipLaunchKernelGGL((flash_ring_kernel<4, 4>), grid64, dim3(256), 4 * 16384, 0, f2); });
    diff("ring (4 stages, 64 queries per wave)");
    {
        const dim3 grid256((T + 255) / 256, H, B);
        hipFuncSetAttribute((const void*)flash_stag_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
        hipFuncSetAttribute((const void*)flash_stag_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
        hipFuncSetAttribute((const void*)flash_stag_kernel<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * 16384);
        hipMemset(o2, 0, nq * 2);
        timeit("staggered SIMD partners, 8 waves, ring of 4, one block per CU", [&] { hipLaunchKernelGGL((flash_stag_kernel<4, 1>), grid256, dim3(512), 4 * 16384, 0, f2); });
        diff("staggered (ring 4, 1 block)");
        hipMemset(o2, 0, nq * 2);
        timeit("staggered SIMD partners, 8 waves, ring of 4, two blocks per CU", [&] { hipLaunchKernelGGL((flash_stag_kernel<4, 2>), grid256, dim3(512), 4 * 16384, 0, f2); });
        diff("staggered (ring 4, 2 blocks)");
        hipMemset(o2, 0, nq * 2);
        timeit("staggered SIMD partners, 8 waves, ring of 5, one block per CU", [&] { hipLaunchKernelGGL((flash_stag_kernel<5, 1>), grid256, dim3(512), 5 * 16384, 0, f2); });
        diff("staggered (ring 5, 1 block)");
    }
    {   // where a wave's time goes (ring, 3 stages, 32 queries per wave), under the real concurrency of the full grid
        const size_t nb = (size_t)grid.x * grid.y * grid.z;
        long long* tr; hipMalloc(&tr, nb * 4 * 5 * 8); hipMemset(tr, 0, nb * 4 * 5 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &tr, sizeof tr);
        hipFuncSetAttribute((const void*)flash_ring_kernel<3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
        timeit("LDS-DMA ring, 3 stages, with the phase clock", [&] { hipLaunchKernelGGL((flash_ring_kernel<3, 2, true>), grid, dim3(256), 3 * 16384, 0, f2); });
        std::vector<long long> ht(nb * 4 * 5);
        hipMemcpy(ht.data(), tr, ht.size() * 8, hipMemcpyDeviceToHost);
        double sum[4] = {0, 0, 0, 0}, tiles = 0;
        for (size_t i = 0; i < nb * 4; ++i) { for (int j = 0; j < 4; ++j) sum[j] += (double)ht[i * 5 + j]; tiles += (double)ht[i * 5 + 4]; }
        printf("    per wave and key tile (s_memtime ticks): wait + barrier %.0f | S = K.Q^T %.0f | softmax %.0f | O += V.P %.0f | total %.0f\n",
               sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, (sum[0] + sum[1] + sum[2] + sum[3]) / tiles);
    }
    return 0;
}
