// flash_enc_kernel: the encoder's self-attention (head dim 64, no mask, not causal: modeling_glmasr.py:171-221 via sdpa) on
// v_mfma_f32_32x32x16_{bf16,f16}.  Round 5 rewrite of the head-dim-64 instance of flash_attn_kernel (attn.hip), which the profile showed
// VALU-bound (0.30 of the MFMA peak: per score element max + fma + exp + add + half a convert, and two waves per SIMD that both queue
// VALU work behind their own MFMAs).  What is different here:
//   * 32x32x16 MFMAs: half the matrix instructions per FLOP (an MFMA holds the SIMD's vector issue for 8 cycles whatever its shape);
//   * 64 queries per wave (two 32-query blocks), 256 per block: every K / V^T fragment read from LDS feeds two MFMAs, and a key tile is
//     staged once per 256 queries instead of once per 128 (half the L2 -> LDS traffic);
//   * K and V^T tiles arrive by LDS-DMA (global_load_lds_dwordx4) into two stages, one barrier per key tile, no staging registers; the
//     XOR swizzle that makes the ds_read_b128 fragment reads conflict-free sits on the SOURCE address;
//   * the running maximum is taken from the FIRST key tile and then left alone: p = exp2((s - m) * c) needs no per-tile maximum, no
//     per-tile rescale of O and l.  Softmax is invariant to the choice of m, and fp32 / bf16 keep their RELATIVE precision at any
//     magnitude, so nothing is lost while p stays finite; a tile whose partial row sum leaves the safe range (a later score more than
//     ~40 / c above m: with c = 0.18 that is 27 nats) is redone through the exact path (maximum, rescale), which is also tile 0's path.
//     Per score element that leaves fma + exp + add + half a v_cvt_pk.
// Formulation as in attn.hip: S^T = K . Q^T and O^T = V^T . P^T, the QUERY on the MFMA column (lane & 31), so the statistics are per-lane
// scalars and the S^T accumulator, converted, IS the B operand of the second product.  With 32x32x16 the accumulator row of register i in
// lane half h is (i & 3) + 8 (i >> 2) + 4 h; K rows are fed in the order kappa (bits 2 and 3 of the row swapped) so that registers
// 8 s .. 8 s + 7 of half h are the 8 CONSECUTIVE keys 16 s + 8 h .. + 7 of the 32-key block: the V^T fragment of k-step s is then one
// aligned 16-byte read.
#include "common.h"
#include "kernels.h"
#include <type_traits>

template <typename T> struct MF32;
template <> struct MF32<bf16_t> {
    static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MF32<f16_t> {
    static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

typedef unsigned u32x2e_t __attribute__((ext_vector_type(2)));
// both lane halves (l and l ^ 32 hold the same query): with equal operands permlane32_swap leaves the own value in one result and the partner's in the other
__device__ __forceinline__ float halves_max(float v) {
    u32x2e_t a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float halves_sum(float v) {
    u32x2e_t a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

#define FE_TILE 8192                      // one K tile (64 keys x 128 B) or one V^T tile (64 hd rows x 128 B)
#define FE_STAGE (2 * FE_TILE)
// A tile's partial row sum at or above FE_BIG (or not finite) sends the tile through the exact path.  bf16 has fp32's exponent range; an fp16
// probability overflows at 65504, so in fp16 (int8 mode / the fp16 test mode) the fixed maximum may lag the true one by at most ~10 nats.
template <typename T> struct FEBig { static constexpr float v = 1.0e30f; };
template <> struct FEBig<f16_t> { static constexpr float v = 3.0e4f; };
#define FE_BIG (FEBig<T>::v)

// MODE bit 0: the second product of the two query blocks as separate MFMA runs (query block A's P.V beside block B's softmax)
//      bit 1: exact path on every tile (the classic online softmax; A/B and the reference for the fast path's tests)
template <typename T, int MODE>
__device__ __forceinline__ void flash_enc_body(const FlashArgs& a, char* smem) {          // smem: 2 * FE_STAGE bytes, 16-byte aligned
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    constexpr bool SPLIT = (MODE & 1) != 0, ALWAYS_EXACT = (MODE & 2) != 0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hq = blockIdx.y, kh = hq / (a.Hq / a.Hkv);
    const int q_len = a.q_len ? a.q_len[b] : a.T;
    const int kv_len = a.kv_len ? a.kv_len[b] : a.T;
    if ((int)blockIdx.x * 256 >= q_len) return;                             // block-uniform
    const int q0 = blockIdx.x * 256 + wid * 64;
    const long qbase = a.q_off ? (long)a.q_off[b] * a.q_ld : (long)b * a.q_seq_stride;
    const long obase = a.q_off ? (long)a.q_off[b] * a.o_ld : (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld;
    const T* Q = (const T*)a.Q + qbase + (long)hq * 64;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)kh * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)kh * a.vt_head_stride;

    // ---- DMA sources.  Per tile and wave: two 1 KiB pieces of K (rows 16 w + 8 i .. + 7) and two of V^T; lane -> (row, 16-byte position)
    // of the LDS image, which holds global chunk pos ^ ((row >> 1) & 7) of that row.
    const T* srcK[2]; const T* srcV[2]; int krow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wid * 16 + i * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
        krow[i] = row;
        srcK[i] = K + ch * 8;
        srcV[i] = Vt + (long)row * a.vt_ld + ch * 8;
    }
    auto dma = [&](int kt, int stage) __attribute__((always_inline)) {
        char* dst = smem + stage * FE_STAGE + wid * 2048;
        const int key0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(key0 + krow[i], kv_len - 1);                 // rows past the sequence are masked below; never read outside it
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcK[i] + (long)key * a.k_ld),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcV[i] + key0),
                                             (__attribute__((address_space(3))) void*)(dst + FE_TILE + i * 1024), 16, 0, 0);
    };

    const int n_tiles = (kv_len + 63) / 64;
    const long bidx = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (a.dbg && tid == 0) { a.dbg[bidx * 4 + 0] = __builtin_amdgcn_s_memtime(); a.dbg[bidx * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
    dma(0, 0);

    // ---- Q fragments (B operand of S^T = K . Q^T): lane holds Q[query = 32 qb + r][hd = 16 ks + 8 h .. + 7]
    V8 qf[2][4];
    int qrow[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qrow[qb] = q0 + qb * 32 + r;
        const int qr = qrow[qb] < q_len ? qrow[qb] : q_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qb][ks] = *(const V8*)(Q + (long)qr * a.q_ld + ks * 16 + h * 8);
    }
    // retire the Q loads here (attn.hip: otherwise the compiler's bookkeeping carries them into the loop as "possibly pending" and waits vmcnt(0)
    // in front of the first MFMAs of every tile, i.e. for the DMA it has just issued)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qb][ks]));

    // ---- fragment read offsets inside a tile.  K: MFMA row r reads key kappa(r) (bits 2 and 3 swapped); V^T: row r of the 32-row block
    const int kap = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);
    int koff[4], voff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        koff[s] = kap * 128 + (((2 * s + h) ^ ((kap >> 1) & 7)) << 4);
        voff[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4);
    }

    f32x16 oacc[2][2];                                                       // [query block][hd block of 32]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb][hb][i] = 0.f;
    float mrun[2] = {-1e30f, -1e30f}, lrun[2] = {0.f, 0.f};
    const float cexp = a.scale * 1.44269504088896341f;

    for (int kt = 0; kt < n_tiles; ++kt) {
        const int stage = kt & 1;
        // tile kt's pieces of THIS wave have landed (they were issued a whole tile ago); the barrier makes every wave's pieces visible and says that
        // every wave is done reading the other stage (tile kt - 1), which tile kt + 1 may now overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < n_tiles) dma(kt + 1, stage ^ 1);
        const char* sK = smem + stage * FE_STAGE;
        const char* sV = sK + FE_TILE;
        const int key0 = kt * 64;

        // ---- S^T = K . Q^T: s[qb][kb] = 32 keys x 32 queries, lane (r, h) register i: query r, key 32 kb + 16 (i >> 3) + 8 h + (i & 7)
        f32x16 s[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[0][kb][i] = 0.f; s[1][kb][i] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const V8 kf = *(const V8*)(sK + kb * 4096 + koff[ks]);
                s[0][kb] = MF32<T>::mfma(kf, qf[0][ks], s[0][kb]);
                s[1][kb] = MF32<T>::mfma(kf, qf[1][ks], s[1][kb]);
            }
        }

        // ---- softmax of one query block -> P^T fragments pf[s4] (k-step s4 = 2 kb + (i >> 3): keys 16 s4 + 8 h .. + 7)
        const bool edge = key0 + 64 > kv_len;
        V8 pf[2][4];
        auto softmax = [&](auto qbc, bool exact) __attribute__((always_inline)) -> float {
            constexpr int qb = decltype(qbc)::value;
            if (edge) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = key0 + kb * 32 + 16 * (i >> 3) + 8 * h + (i & 7);
                        s[qb][kb][i] = key < kv_len ? s[qb][kb][i] : -1e30f;
                    }
            }
            if (exact) {
                float mx = -1e30f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[qb][kb][i]);
                mx = halves_max(mx);
                const float mnew = fmaxf(mrun[qb], mx);
                const float alpha = __builtin_amdgcn_exp2f((mrun[qb] - mnew) * cexp);
                mrun[qb] = mnew;
                lrun[qb] *= alpha;
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) oacc[qb][hb][i] *= alpha;
                }
            }
            const float moff = -mrun[qb] * cexp;
            // one accumulator chain: with two, hipcc SLP-packs the adds into v_pk_add_f32, which costs more issue time than the two v_add_f32 it replaces
            float ps = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kb][i], cexp, moff));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][kb][i + 1], cexp, moff));
                    ps += p0; ps += p1;
                    pf[qb][kb * 2 + (i >> 3)][i & 7] = (T)p0;
                    pf[qb][kb * 2 + (i >> 3)][(i & 7) + 1] = (T)p1;
                }
            return ps;
        };
        auto run_softmax = [&](auto qbc) __attribute__((always_inline)) {
            constexpr int qb = decltype(qbc)::value;
            const bool first = ALWAYS_EXACT || kt == 0;
            float pt = softmax(qbc, first);
            if (!first && __builtin_amdgcn_ballot_w64(!(pt < FE_BIG)) != 0) pt = softmax(qbc, true);   // rare: a score far above the first tile's maximum
            lrun[qb] += pt;
        };

        if (!SPLIT) {
            run_softmax(std::integral_constant<int, 0>{});
            run_softmax(std::integral_constant<int, 1>{});
            // ---- O^T += V^T . P^T
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const V8 vf = *(const V8*)(sV + hb * 4096 + voff[s4]);
                    oacc[0][hb] = MF32<T>::mfma(vf, pf[0][s4], oacc[0][hb]);
                    oacc[1][hb] = MF32<T>::mfma(vf, pf[1][s4], oacc[1][hb]);
                }
        } else {
            run_softmax(std::integral_constant<int, 0>{});
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const V8 vf = *(const V8*)(sV + hb * 4096 + voff[s4]);
                    oacc[0][hb] = MF32<T>::mfma(vf, pf[0][s4], oacc[0][hb]);
                }
            run_softmax(std::integral_constant<int, 1>{});
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const V8 vf = *(const V8*)(sV + hb * 4096 + voff[s4]);
                    oacc[1][hb] = MF32<T>::mfma(vf, pf[1][s4], oacc[1][hb]);
                }
        }
    }

    // ---- epilogue: O[query][64 hq + hd] = O^T / l.  Register i of hd block hb: hd = 32 hb + 8 (i >> 2) + 4 h + (i & 3); the two lane halves hold the two
    // 8-byte halves of each 16-byte piece, exchanged with permlane32_swap so that every lane stores 16 bytes (guide T21)
    T* O = (T*)a.O + obase + (long)hq * 64;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l = halves_sum(lrun[qb]);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                V4 oa, ob;
#pragma unroll
                for (int j = 0; j < 4; ++j) { oa[j] = (T)(oacc[qb][hb][4 * g + j] / l); ob[j] = (T)(oacc[qb][hb][4 * g + 4 + j] / l); }
                unsigned ax = ((const unsigned*)&oa)[0], ay = ((const unsigned*)&oa)[1], bx = ((const unsigned*)&ob)[0], by = ((const unsigned*)&ob)[1];
                u32x2e_t sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                u32x2e_t sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                // lower half: [own group g | upper's group g] = hd 8 g .. 8 g + 7; upper half: [lower's group g + 1 | own group g + 1] = hd 8 (g + 1) ..
                const uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                if (qrow[qb] < q_len) *(uint4*)(O + (long)qrow[qb] * a.o_ld + hb * 32 + 8 * (g + h)) = v;
            }
    }
    if (a.dbg && tid == 0) { a.dbg[bidx * 4 + 2] = __builtin_amdgcn_s_memtime(); a.dbg[bidx * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
}


template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void flash_enc_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * FE_STAGE];        // the kernel's only LDS object
    flash_enc_body<T, MODE>(a, smem);
}

typedef unsigned u32x4e_t __attribute__((ext_vector_type(4)));
// ---- flash_encp_kernel: the same arithmetic as flash_enc_kernel, software-pipelined and issued in a fixed order.
// Why: a SIMD has ONE vector issue port for its waves, and an MFMA that waits for the matrix pipe blocks the VALU work queued behind it in its
// wave.  flash_enc_kernel's waves alternate a pure-MFMA phase with a pure-VALU phase; two such waves on a SIMD fall into step and the two pipes
// take turns (measured: the kernel spends the SUM of its MFMA and VALU time: 2 570 SIMD cycles per wave and key tile for 1 024 cycles of MFMA and
// ~1 150 of VALU issue).  Here a wave owns its SIMD (256 threads per block, one block per CU, up to 512 registers) and its two 32-query blocks run
// half a key tile apart, so that every MFMA has independent softmax work to issue behind it:
//     slot 1:  softmax of block A, tile t      beside   O_B += V(t-1) . P_B(t-1)  and  S_B(t)   = K(t)   . Q_B      (16 MFMAs)
//     slot 2:  softmax of block B, tile t      beside   O_A += V(t)   . P_A(t)    and  S_A(t+1) = K(t+1) . Q_A      (16 MFMAs)
// A slot is 16 groups, each ONE asm statement - so the issue order is the order written -: [v_mfma_f32_32x32x16] [2 fma] [2 exp] [2 add]
// [1 packed convert].  The MFMA holds the issue port for 8 of its 32 matrix-pipe cycles; the seven VALU instructions take ~36: the port is
// never idle and the pipe never waits for more than the VALU surplus (the guide's forward-attention recipe, with the compiler still doing the
// register allocation, the LDS fragment reads two groups ahead and their waits).
// Registers: O (64) and the Q fragments (32) are touched by MFMAs only and live in AGPRs ("a" operands: an MFMA takes A / B / C from either
// file); S, P and everything the VALU touches are VGPRs.  O is never rescaled: the running maximum is the first tile's (see the header), and a
// block in which some partial row sum left the safe range is recomputed by the classic online softmax after the loop (exact_block; never seen on
// real activations, forced in tests/test_gpu_flash_enc.py).
// Four LDS stages (V(t-1) | K(t), V(t) | K(t+1) | the tile in flight), one barrier per key tile.
template <typename T, bool AG> struct FEAsm;       // AG: O and the Q fragments are AGPR operands (one wave per SIMD) or plain VGPRs (two)
// The softmax beside an MFMA is itself pipelined over three groups, so that no instruction reads a result younger than a whole group (a
// v_fma -> v_exp -> v_add chain inside ONE group waits for each stage's latency: measured 109 cycles per group instead of the ~45 the issue costs add up to):
//     group g:   add + add + packed convert of group g-1's exponentials (x0, x1)  |  exp of group g's scaled scores (t0, t1, made in group g-1)  |
//                fma (scale, offset) of group g+1's scores
// x0 / x1 / t0 / t1 are carried from statement to statement ("+v").
#define FE_ADDCVT(CVT) "v_add_f32 %[ps], %[ps], %[x0]\n\tv_add_f32 %[ps], %[ps], %[x1]\n\t" CVT " %[pk], %[x0], %[x1]\n\t"
#define FE_EXP "v_exp_f32 %[x0], %[t0]\n\tv_exp_f32 %[x1], %[t1]\n\t"
#define FE_FMA "v_fma_f32 %[t0], %[e0], %[c], %[mo]\n\tv_fma_f32 %[t1], %[e1], %[c], %[mo]"
// RD: the statement opens with the LDS read of the fragment two groups on (early-clobber output: its register must not be one this statement's
// MFMA reads) and waits with a COUNTED lgkmcnt for its own fragment, which an earlier statement requested.  The compiler does not know that
// fragment registers are in flight: it only keeps the statements in order (volatile) and the registers allocated.  Left to wait itself (C++
// ds_reads feeding asm operands) it drains lgkmcnt(0) every third group - the read just issued included.
#define FE_RD "ds_read_b128 %[nf], %[na] offset:%c[off]\n\ts_waitcnt lgkmcnt(%c[w])\n\t"
#define FE_NORD "s_waitcnt lgkmcnt(%c[w])\n\t"
#define FE_CARRY [x0] "+v"(x0), [x1] "+v"(x1), [t0] "+v"(t0), [t1] "+v"(t1)
#define FE_ASM_IMPL(TY, AGB, OC, MF, CVT)                                                                                                                   \
    template <> struct FEAsm<TY, AGB> {                                                                                                                           \
        typedef typename ET<TY>::v8 V8;                                                                                                                      \
        static __device__ __forceinline__ void lds(V8& f, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(addr)); }                      \
        /* group 0: O (AGPRs) += fr . p; exp of group 0, fma of group 1 */                                                                                     \
        template <int W, int OFF> static __device__ __forceinline__ void pv_first(f32x16& o, V8 fr, u32x4e_t b, float e0, float e1, float c, float moff,     \
                                                                                  float& x0, float& x1, float& t0, float& t1, V8& nf, unsigned na) {          \
            asm volatile(FE_RD MF " %[o], %[fr], %[b], %[o]\n\t" FE_EXP FE_FMA                                                                               \
                         : [o] "+" OC(o), [x0] "=&v"(x0), [x1] "=&v"(x1), [t0] "+v"(t0), [t1] "+v"(t1), [nf] "=&v"(nf)                                         \
                         : [fr] "v"(fr), [b] "v"(b), [e0] "v"(e0), [e1] "v"(e1), [c] "v"(c), [mo] "v"(moff), [na] "v"(na), [off] "n"(OFF), [w] "n"(W));       \
        }                                                                                                                                                    \
        template <int W, int OFF> static __device__ __forceinline__ void pv_mid(f32x16& o, V8 fr, u32x4e_t b, float e0, float e1, float c, float moff,       \
                                                                                float& ps, unsigned& pk, float& x0, float& x1, float& t0, float& t1, V8& nf, unsigned na) { \
            asm volatile(FE_RD MF " %[o], %[fr], %[b], %[o]\n\t" FE_ADDCVT(CVT) FE_EXP FE_FMA                                                                \
                         : [o] "+" OC(o), [ps] "+v"(ps), [pk] "=&v"(pk), FE_CARRY, [nf] "=&v"(nf)                                                              \
                         : [fr] "v"(fr), [b] "v"(b), [e0] "v"(e0), [e1] "v"(e1), [c] "v"(c), [mo] "v"(moff), [na] "v"(na), [off] "n"(OFF), [w] "n"(W));       \
        }                                                                                                                                                    \
        /* S (VGPRs) = fr . q (AGPRs) / += */                                                                                                                 \
        template <int W, int OFF> static __device__ __forceinline__ void qk0_mid(f32x16& o, V8 fr, V8 b, float e0, float e1, float c, float moff,            \
                                                                                 float& ps, unsigned& pk, float& x0, float& x1, float& t0, float& t1, V8& nf, unsigned na) { \
            asm volatile(FE_RD MF " %[o], %[fr], %[b], 0\n\t" FE_ADDCVT(CVT) FE_EXP FE_FMA                                                                   \
                         : [o] "=&v"(o), [ps] "+v"(ps), [pk] "=&v"(pk), FE_CARRY, [nf] "=&v"(nf)                                                             \
                         : [fr] "v"(fr), [b] OC(b), [e0] "v"(e0), [e1] "v"(e1), [c] "v"(c), [mo] "v"(moff), [na] "v"(na), [off] "n"(OFF), [w] "n"(W));       \
        }                                                                                                                                                    \
        template <int W, int OFF> static __device__ __forceinline__ void qk_mid(f32x16& o, V8 fr, V8 b, float e0, float e1, float c, float moff,             \
                                                                                float& ps, unsigned& pk, float& x0, float& x1, float& t0, float& t1, V8& nf, unsigned na) { \
            asm volatile(FE_RD MF " %[o], %[fr], %[b], %[o]\n\t" FE_ADDCVT(CVT) FE_EXP FE_FMA                                                                \
                         : [o] "+v"(o), [ps] "+v"(ps), [pk] "=&v"(pk), FE_CARRY, [nf] "=&v"(nf)                                                              \
                         : [fr] "v"(fr), [b] OC(b), [e0] "v"(e0), [e1] "v"(e1), [c] "v"(c), [mo] "v"(moff), [na] "v"(na), [off] "n"(OFF), [w] "n"(W));       \
        }                                                                                                                                                    \
        template <int W> static __device__ __forceinline__ void qk_mid_nord(f32x16& o, V8 fr, V8 b, float e0, float e1, float c, float moff,                 \
                                                                            float& ps, unsigned& pk, float& x0, float& x1, float& t0, float& t1) {            \
            asm volatile(FE_NORD MF " %[o], %[fr], %[b], %[o]\n\t" FE_ADDCVT(CVT) FE_EXP FE_FMA                                                              \
                         : [o] "+v"(o), [ps] "+v"(ps), [pk] "=&v"(pk), FE_CARRY                                                                              \
                         : [fr] "v"(fr), [b] OC(b), [e0] "v"(e0), [e1] "v"(e1), [c] "v"(c), [mo] "v"(moff), [w] "n"(W));                                    \
        }                                                                                                                                                    \
        /* group 15: no further scores to scale; the trailing s_nop is the trans -> VALU wait state for the compiler's code behind the statement */            \
        template <int W> static __device__ __forceinline__ void qk_last(f32x16& o, V8 fr, V8 b, float& ps, unsigned& pk, float& x0, float& x1, float t0, float t1) { \
            asm volatile(FE_NORD MF " %[o], %[fr], %[b], %[o]\n\t" FE_ADDCVT(CVT) "v_exp_f32 %[x0], %[t0]\n\tv_exp_f32 %[x1], %[t1]\n\ts_nop 1"             \
                         : [o] "+v"(o), [ps] "+v"(ps), [pk] "=&v"(pk), [x0] "+v"(x0), [x1] "+v"(x1)                                                          \
                         : [fr] "v"(fr), [b] OC(b), [t0] "v"(t0), [t1] "v"(t1), [w] "n"(W));                                                                \
        }                                                                                                                                                    \
        /* the MFMAs alone (prologue, last P.V) */                                                                                                            \
        static __device__ __forceinline__ void pv(f32x16& o, V8 fr, u32x4e_t b) { asm volatile(MF " %0, %1, %2, %0" : "+" OC(o) : "v"(fr), "v"(b)); }            \
        static __device__ __forceinline__ void qk(f32x16& o, V8 fr, V8 b) { asm volatile(MF " %0, %1, %2, %0" : "+v"(o) : "v"(fr), OC(b)); }                 \
        static __device__ __forceinline__ void qk0(f32x16& o, V8 fr, V8 b) { asm volatile(MF " %0, %1, %2, 0" : "=&v"(o) : "v"(fr), OC(b)); }                \
        static __device__ __forceinline__ void zero(f32x16& o, V8 z) { asm volatile(MF " %0, %1, %1, 0" : "=&" OC(o) : "v"(z)); }                               \
    };
FE_ASM_IMPL(bf16_t, true, "a", "v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32")
FE_ASM_IMPL(f16_t, true, "a", "v_mfma_f32_32x32x16_f16", "v_cvt_pk_f16_f32")
FE_ASM_IMPL(bf16_t, false, "v", "v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32")
FE_ASM_IMPL(f16_t, false, "v", "v_mfma_f32_32x32x16_f16", "v_cvt_pk_f16_f32")

template <typename T, bool AG>
__global__ __launch_bounds__(256, AG ? 1 : 2) void flash_encp_kernel(FlashArgs a) {
    typedef typename ET<T>::v8 V8;
    typedef typename ET<T>::v4 V4;
    typedef FEAsm<T, AG> AS;
    __shared__ __attribute__((aligned(16))) char smem[4 * FE_STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, hq = blockIdx.y, kh = hq / (a.Hq / a.Hkv);
    const int q_len = a.q_len ? a.q_len[b] : a.T;
    const int kv_len = a.kv_len ? a.kv_len[b] : a.T;
    if ((int)blockIdx.x * 256 >= q_len) return;
    const int q0 = blockIdx.x * 256 + wid * 64;
    const long qbase = a.q_off ? (long)a.q_off[b] * a.q_ld : (long)b * a.q_seq_stride;
    const long obase = a.q_off ? (long)a.q_off[b] * a.o_ld : (long)b * (a.q_seq_stride / a.q_ld) * a.o_ld;
    const T* Q = (const T*)a.Q + qbase + (long)hq * 64;
    const T* K = (const T*)a.K + (long)b * a.k_seq_stride + (long)kh * a.k_head_stride;
    const T* Vt = (const T*)a.Vt + (long)b * a.vt_seq_stride + (long)kh * a.vt_head_stride;
    T* O = (T*)a.O + obase + (long)hq * 64;

    const T* srcK[2]; const T* srcV[2]; int krow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wid * 16 + i * 8 + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
        krow[i] = row;
        srcK[i] = K + ch * 8;
        srcV[i] = Vt + (long)row * a.vt_ld + ch * 8;
    }
    auto dma = [&](int kt) __attribute__((always_inline)) {
        char* dst = smem + (kt & 3) * FE_STAGE + wid * 2048;
        const int key0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = min(key0 + krow[i], kv_len - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcK[i] + (long)key * a.k_ld),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcV[i] + key0),
                                             (__attribute__((address_space(3))) void*)(dst + FE_TILE + i * 1024), 16, 0, 0);
    };
    const int n_tiles = (kv_len + 63) / 64;
    const long bidx = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (a.dbg && tid == 0) { a.dbg[bidx * 4 + 0] = __builtin_amdgcn_s_memtime(); a.dbg[bidx * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
    dma(0);
    if (n_tiles > 1) dma(1);

    const int qrowA = q0 + r, qrowB = q0 + 32 + r;
    V8 qA[4], qB[4];
    {
        const int ra = qrowA < q_len ? qrowA : q_len - 1, rb = qrowB < q_len ? qrowB : q_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qA[ks] = *(const V8*)(Q + (long)ra * a.q_ld + ks * 16 + h * 8); qB[ks] = *(const V8*)(Q + (long)rb * a.q_ld + ks * 16 + h * 8); }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                                          // retired here (and, AG, at home in AGPRs)
        if constexpr (AG) { asm volatile("" : "+a"(qA[ks])); asm volatile("" : "+a"(qB[ks])); }
        else { asm volatile("" : "+v"(qA[ks])); asm volatile("" : "+v"(qB[ks])); }
    }

    const int kap = (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1);
    int koff[4], voff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        koff[s] = kap * 128 + (((2 * s + h) ^ ((kap >> 1) & 7)) << 4);
        voff[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) << 4);
    }
    f32x16 oA[2], oB[2], sA[2], sB[2];
    u32x4e_t pA[4], pB[4];
    {
        V8 z;
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = (T)0.f;
        AS::zero(oA[0], z); AS::zero(oA[1], z); AS::zero(oB[0], z); AS::zero(oB[1], z);
    }
    const float cexp = a.scale * 1.44269504088896341f;
    float lA = 0.f, lB = 0.f, bad = 0.f;

    auto kfrag = [&](int kt, int g) __attribute__((always_inline)) -> V8 { return *(const V8*)(smem + (kt & 3) * FE_STAGE + (g >> 2) * 4096 + koff[g & 3]); };
    auto vfrag = [&](int kt, int g) __attribute__((always_inline)) -> V8 { return *(const V8*)(smem + (kt & 3) * FE_STAGE + FE_TILE + (g >> 2) * 4096 + voff[g & 3]); };
    // S(kt) = K(kt) . Q without a softmax beside it (prologue)
    auto qk_only = [&](int kt, const V8 (&q)[4], f32x16 (&s)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { if ((g & 3) == 0) AS::qk0(s[g >> 2], kfrag(kt, g), q[0]); else AS::qk(s[g >> 2], kfrag(kt, g), q[g & 3]); }
    };
    auto row_max = [&](const f32x16 (&s)[2], int kt) __attribute__((always_inline)) -> float {
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ok = (kt * 64 + kb * 32 + 16 * (i >> 3) + 8 * h + (i & 7)) < kv_len;
                mx = fmaxf(mx, ok ? s[kb][i] : -1e30f);
            }
        return halves_max(mx);
    };
    auto mask_tail = [&](f32x16 (&s)[2], int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = (kt * 64 + kb * 32 + 16 * (i >> 3) + 8 * h + (i & 7)) < kv_len ? s[kb][i] : -1e30f;
    };
    auto sm_only = [&](const f32x16 (&s)[2], u32x4e_t (&p)[4], float moff) __attribute__((always_inline)) -> float {
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i], cexp, moff));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][i + 1], cexp, moff));
                ps += p0; ps += p1;
                typename ET<T>::v2 pr; pr[0] = (T)p0; pr[1] = (T)p1;
                p[kb * 2 + (i >> 3)][(i & 7) >> 1] = __builtin_bit_cast(unsigned, pr);
            }
        return ps;
    };
    // one slot: groups 0-7  O += V(kt_v) . p_in,  groups 8-15  s_out = K(kt_k) . q;  beside them the softmax of s_in -> p_out
    auto slot = [&](int kt_v, const u32x4e_t (&p_in)[4], f32x16 (&o)[2], int kt_k, const V8 (&q)[4], f32x16 (&s_out)[2],
                    const f32x16 (&s_in)[2], u32x4e_t (&p_out)[4], float moff) __attribute__((always_inline)) -> float {
        // LDS byte addresses of the fragments: group g < 8 reads V(kt_v) at vbase[g & 3] + 4096 (g >> 2), g >= 8 reads K(kt_k) at kbase[g & 3] + 4096 ((g - 8) >> 2)
        const unsigned sv = (unsigned)(size_t)(smem) + (kt_v & 3) * FE_STAGE + FE_TILE, sk = (unsigned)(size_t)(smem) + (kt_k & 3) * FE_STAGE;
        unsigned vb[4], kb_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { vb[i] = sv + voff[i]; kb_[i] = sk + koff[i]; }
        V8 fr[3];
        AS::lds(fr[0], vb[0]); AS::lds(fr[1], vb[1]);
        float ps = 0.f, x0, x1;
        float t0 = __builtin_fmaf(s_in[0][0], cexp, moff), t1 = __builtin_fmaf(s_in[0][1], cexp, moff);       // group 0's scaled scores
        // group g's statement converts group g-1's pair: P position of pair j = (key block j >> 3, element 2 (j & 7))
        auto put = [&](int j, unsigned pk) __attribute__((always_inline)) { p_out[(j >> 3) * 2 + ((2 * (j & 7)) >> 3)][((2 * (j & 7)) & 7) >> 1] = pk; };
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int gn = g + 1 < 16 ? g + 1 : 15;                          // the group whose scores this statement scales
            const float e0 = s_in[gn >> 3][2 * (gn & 7)], e1 = s_in[gn >> 3][2 * (gn & 7) + 1];
            unsigned pk = 0;
            const int n = g + 2;                                              // the fragment this group requests
            const unsigned na = n < 8 ? vb[n & 3] : kb_[n & 3];
            V8& nf = fr[n % 3];
            V8 cur = fr[g % 3];
            if (g == 0) AS::template pv_first<2, 0>(o[0], cur, p_in[0], e0, e1, cexp, moff, x0, x1, t0, t1, nf, na);
            else if (g < 2) AS::template pv_mid<2, 0>(o[0], cur, p_in[g & 3], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g < 6) AS::template pv_mid<2, 4096>(o[g >> 2], cur, p_in[g & 3], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g < 8) AS::template pv_mid<2, 0>(o[1], cur, p_in[g & 3], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g == 8) AS::template qk0_mid<2, 0>(s_out[0], cur, q[0], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g == 12) AS::template qk0_mid<2, 4096>(s_out[1], cur, q[0], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g < 10) AS::template qk_mid<2, 0>(s_out[0], cur, q[g & 3], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g < 14) AS::template qk_mid<2, 4096>(s_out[(g - 8) >> 2], cur, q[g & 3], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1, nf, na);
            else if (g == 14) AS::template qk_mid_nord<1>(s_out[1], cur, q[2], e0, e1, cexp, moff, ps, pk, x0, x1, t0, t1);
            else AS::template qk_last<0>(s_out[1], cur, q[3], ps, pk, x0, x1, t0, t1);
            if (g > 0) put(g - 1, pk);
            __builtin_amdgcn_sched_barrier(0);
        }
        ps += x0; ps += x1;
        {
            typename ET<T>::v2 pr; pr[0] = (T)x0; pr[1] = (T)x1;
            put(15, __builtin_bit_cast(unsigned, pr));
        }
        return ps;
    };

    // ---- prologue: tile 0's scores of both query blocks, their maxima (the exponent offset of the whole row)
    if (n_tiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    qk_only(0, qA, sA);
    qk_only(0, qB, sB);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                        // (MFMA results written inside asm statements: their wait states, by hand)
    const float moffA = -row_max(sA, 0) * cexp, moffB = -row_max(sB, 0) * cexp;

    auto top = [&](int kt) __attribute__((always_inline)) {
        // tile kt + 1 has landed for every wave; every wave has left iteration kt - 1, so the stage of tile kt - 2 may take tile kt + 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < n_tiles) dma(kt + 2);
    };
    const bool ragged = (kv_len & 63) != 0;
    // iteration 0 (peeled: there is no P_B(-1), S_B(0) exists already; O_A / O_B are touched by asm statements only from here to the store)
    top(0);
    if (n_tiles == 1 && ragged) mask_tail(sA, 0);
    {
        float ps = sm_only(sA, pA, moffA);
        lA += ps; bad = fmaxf(bad, ps < FE_BIG ? 0.f : 1.f);
        if (n_tiles == 1 && ragged) mask_tail(sB, 0);
        ps = slot(0, pA, oA, 1, qA, sA, sB, pB, moffB);
        lB += ps; bad = fmaxf(bad, ps < FE_BIG ? 0.f : 1.f);
    }
    for (int kt = 1; kt < n_tiles; ++kt) {
        top(kt);
        const bool last = kt + 1 == n_tiles;
        if (last && ragged) { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); mask_tail(sA, kt); }
        float ps = slot(kt - 1, pB, oB, kt, qB, sB, sA, pA, moffA);
        lA += ps; bad = fmaxf(bad, ps < FE_BIG ? 0.f : 1.f);
        if (last && ragged) { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); mask_tail(sB, kt); }
        // (last tile: the slot's S_A(kt + 1) is computed from a stale LDS stage and never used)
        ps = slot(kt, pA, oA, kt + 1, qA, sA, sB, pB, moffB);
        lB += ps; bad = fmaxf(bad, ps < FE_BIG ? 0.f : 1.f);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) AS::pv(oB[g >> 2], vfrag(n_tiles - 1, g), pB[g & 3]);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    auto store = [&](const f32x16 (&o)[2], float lr, int qrow) __attribute__((always_inline)) {
        const float l = halves_sum(lr);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                V4 oa, ob;
#pragma unroll
                for (int j = 0; j < 4; ++j) { oa[j] = (T)(o[hb][4 * g + j] / l); ob[j] = (T)(o[hb][4 * g + 4 + j] / l); }
                unsigned ax = ((const unsigned*)&oa)[0], ay = ((const unsigned*)&oa)[1], bx = ((const unsigned*)&ob)[0], by = ((const unsigned*)&ob)[1];
                u32x2e_t sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                u32x2e_t sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                const uint4 v = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                if (qrow < q_len) *(uint4*)(O + (long)qrow * a.o_ld + hb * 32 + 8 * (g + h)) = v;
            }
    };
    // a block in which some partial row sum left the safe range (a score more than ~100 / c above the first tile's maximum: never on real activations)
    // is recomputed with a running maximum; the decision is block-wide because the recomputation shares LDS tiles and barriers
    if (__syncthreads_or(bad != 0.f)) {
        flash_enc_body<T, 2>(a, smem);
        return;
    }
    store(oA, lA, qrowA);
    store(oB, lB, qrowB);
    if (a.dbg && tid == 0) { a.dbg[bidx * 4 + 2] = __builtin_amdgcn_s_memtime(); a.dbg[bidx * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
}

void launch_flash_enc(const FlashArgs& a, int B, int max_q, int mode, hipStream_t s) {
    dim3 grid((max_q + 255) / 256, a.Hq, B), block(256);
    DT_SWITCH(a.dt, T, {
        switch (mode & 7) {
            case 4: hipLaunchKernelGGL((flash_encp_kernel<T, true>), grid, block, 0, s, a); break;     // one wave per SIMD, O / Q in AGPRs
            case 5: hipLaunchKernelGGL((flash_encp_kernel<T, false>), grid, block, 0, s, a); break;    // two waves per SIMD, everything in VGPRs
            case 1: hipLaunchKernelGGL((flash_enc_kernel<T, 1>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((flash_enc_kernel<T, 2>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((flash_enc_kernel<T, 3>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((flash_enc_kernel<T, 0>), grid, block, 0, s, a); break;
        }
    });
}
