// Launch descriptors and host-callable launchers of the HIP kernels (attn.hip, elementwise.hip, logmel.hip).
#pragma once
#include "common.h"

// ---- int8 mode (LLM.int8, asr.py:169-210) plumbing shared by the decode-step kernels ---------------------------------------------
// How a consumer turns the int32 partial slabs of a quantised skinny GEMM into the module's fp16 output:
//   y = fp16(float(sum_ks C32) * (SCA[row] * SCB[col] * 1/127^2)); outlier columns of the row's group are added from the unquantised
//   input and the dequantised int8 weights, y = fp16(y + sum_k x[row][k] * fp16(CB[col][k] * SCB[col] / 127))   (sonic_oracle.c linear_int8)
struct DeqInfo {
    const float* sca;                 // [M] row absmax of the GEMM's quantised input; NULL = plain mode (slabs hold fp32 partial sums)
    const float* scb;                 // [N] row absmax of the weights
    const int8_t* cb; int K;          // [N][K] row-major int8 weights
    const int8_t* cbt;                // optional fragment-tiled copy (launch_tile_weights_i8): a column gather W[:, k] touches one 64-byte
                                      // span per 16 rows there instead of one cache line per row
    const int8_t* cbk; int N;         // optional k-major copy [K][N] (launch_transpose_i8): the gather of 8 consecutive output columns is ONE 8-byte load
    const bf16_t* x16; long ldx16;    // the GEMM's unquantised input (fp16 storage)
    const int* oc_cnt; const int* oc_list; int oc_ld;   // outlier columns per group: count [G], ascending list [G][oc_ld]
    const int* row_group; int group_div;                 // group of row r = row_group ? row_group[r / group_div] : r / group_div
    const float* oc_val;              // optional (decode: one row = one group): x value of every list entry, same indexing as oc_list
    const int* scan_cnt;              // with scan: optional [M][4] per-block counts of the elements >= 6.0 (written beside the partial maxima): all zero - the
                                      // common case - and the consumer skips its scan of the row
    int dbg;                          // timing experiments only (option i8_dbg; results are wrong): bit 0 skip the outlier stage, bit 1 skip the re-quantisation
    int scan;                         // decode consumers that own a row: 1 = nobody listed the row's outliers (its producer did not own whole rows and
                                      // the projection quantised it on the fly, SkinnyArgs.x_amax): the consumer finds them in x16 itself, ascending;
                                      // sca then holds 4 partial maxima per row ([M][4], the row absmax is their maximum).
                                      // oc_list / oc_val are this row's spill space for lists longer than the LDS stage
};
// A producer that owns whole rows also emits them quantised for the next Linear8bitLt (decode step: one row = one reference call,
// so the outlier "columns" are the row's own elements >= 6.0)
struct QuantOut {
    int8_t* q; long ldq;              // [M][K] int8; NULL = off
    float* sca;                       // [M]
    int* oc_cnt; int* oc_list; int oc_ld;
    float* oc_val;                    // optional: the outliers' values, [M][oc_ld] beside oc_list
};

struct FlashArgs {
    const bf16_t* Q; long q_ld;            // token-major, head h at column h*HD
    const bf16_t* K; long k_ld;            // key-major rows
    const bf16_t* Vt; long vt_ld;          // V^T rows: [hd][key]
    bf16_t* O; long o_ld;
    long q_seq_stride, k_seq_stride, k_head_stride, vt_seq_stride, vt_head_stride;
    const int* q_off;                      // optional [B]: first packed token of each sequence (ragged); else b * q_seq_stride
    const int* q_len;                      // optional [B]; else T
    const int* kv_len;                     // optional [B]; else T
    int T, Hq, Hkv;
    float scale;
    int dt;                                // DT_BF16 / DT_F16
    long long* dbg;                        // diagnostics (tools/flash_enc_bench): per block {shader clock, 100 MHz clock} at entry and exit; null in production
};
void launch_flash(const FlashArgs& a, int hd, bool causal, int B, int max_q, hipStream_t s);
void launch_flash_enc(const FlashArgs& a, int B, int max_q, int mode, hipStream_t s);   // attn_enc.hip: head dim 64, no mask (the encoder)

struct DecodeAttnArgs {
    const bf16_t* Q;      // [B][Hq*128] (roped); used when P == null
    const float* P; int ksplit, mpad; const float* cs;   // fused mode: QKV slabs [ks][mpad][(Hq+2Hkv)*128] + rope table [ctx][128]
    bf16_t* Kc;           // [B][Hkv][ctx_max][128]
    bf16_t* Vc;           // [B][Hkv][ctx_max][128]
    bf16_t* O;            // [B][Hq*128]
    const int* kv_len;    // [B] keys visible (new token included)
    int Hq, Hkv, ctx_max;
    float scale;
    int dt;
    DeqInfo dq;           // int8 mode: P holds int32 slabs of the quantised QKV projection
    float* amax_out;      // int8 mode, optional: [B][4] per-block partials (kv head) of the output rows' absmax without the elements >= 6.0; Hkv <= 4,
                          // unused partials stay 0
    int* big_out;         // with amax_out: [B][4] how many of the block's outputs are >= 6.0
    long long* kt;        // diagnostics: per-block timestamps (common.h KT)
    PrefetchRange pf[2];  // experiment (option decode_prefetch): blocks blockIdx.y >= Hkv stream these ranges (weights of later kernels) and exit
    int pf_y;             // ... that many extra grid rows
};
void launch_decode_attn(const DecodeAttnArgs& a, int B, hipStream_t s);

struct RopeAppendArgs {
    const bf16_t* qkv; long ld;          // prefill source [tok][Hq*hd + 2*Hkv*hd]
    const float* P; int ksplit, mpad;    // decode source slabs [ks][mpad][N]
    bf16_t* q_out;                       // [tok][Hq*hd]
    bf16_t* Kc; bf16_t* Vc;              // [B][Hkv][ctx_max][hd]
    bf16_t* Vt; long vt_ld;              // optional [B][Hkv][hd][vt_ld]
    const int* tok_seq; const int* tok_pos;  // per token: sequence index, absolute position
    const float* cs;                     // [ctx_max][hd]: cos[0..hd/2) | sin[0..hd/2)
    int Hq, Hkv, ctx_max, n_tok;
    int dt;
    // prefill, round 5 (optional): packed-token offset and length of every sequence and the longest prompt.  With them the kernel works on tiles of
    // 16 consecutive positions of ONE sequence, so the V^T scratch is written in 16-byte rows instead of one 2-byte store per element
    const int* q_off; const int* q_len; int n_seq, max_p;
};
void launch_rope_append(const RopeAppendArgs& a, bool slab, hipStream_t s);

struct GreedyArgs {
    const float* logits;   // slabs [ksplit][mpad][V] fp32 partial accumulators of the lm_head
    int ksplit, mpad;
    int V, B;
    const bf16_t* table;   // embedding table (tied lm_head), for the next step's input row
    bf16_t* x; int d;      // [B][d] next-step hidden input
    int* out_ids; int out_ld;   // [B][out_ld] generated ids
    int* n_new;            // [B]
    int* finished;         // [B]
    int* kv_len;           // [B] keys visible to the *next* step (incremented here)
    int* tok_pos;          // [B] position of the next token
    const int* max_new;    // [B]
    int* n_active;         // [1] rows still running (host polls)
    const int* dev_err;    // optional: device error word; non-zero turns n_active negative (the host fails the batch)
    int eos[8]; int n_eos; int pad_id;
    float* logits_dump; long dump_stride_step; int* step_counter;  // optional: bf16-rounded logits per step [step][B][V]; counter per row [B]
    const float* norm_w; float norm_eps; bf16_t* y;   // optional: y[B][d] = RMSNorm(x) with the first decoder layer's input norm
    const int* force_ids; int force_ld;                // optional teacher forcing: token n of row b is force_ids[b * force_ld + n] (oracle force_ids)
    int dt;
    QuantOut qo;                                       // int8 mode: y is also emitted quantised (input of layer 0's q/k/v Linear8bitLt)
};
void launch_greedy(const GreedyArgs& a, hipStream_t s);

struct LogmelConst {
    const float* win;      // [400]
    const float* cos_t;    // [400]
    const float* sin_t;    // [400]
    const int* mel_lo;     // [n_mels] first bin of each filter
    const int* mel_cnt;    // [n_mels] taps
    const int* mel_off;    // [n_mels] offset into mel_w
    const float* mel_w;    // packed taps
};
void launch_logmel(const int16_t* pcm, long pcm_stride, const int* n_samples_dev, int max_samples, const LogmelConst& lc,
                   float* logspec, int* segmax, int B, int n_frames, int n_mels, bf16_t* feats_fm, float* feats_f32, hipStream_t s, int dt = DT_BF16);

struct QuantActArgs;
void launch_layernorm(const bf16_t* x, const float* w, const float* b, bf16_t* y, int rows, int d, float eps, hipStream_t s, int dt = DT_BF16,
                      const QuantActArgs* qa = nullptr);
void launch_rmsnorm(const bf16_t* x, const float* w, bf16_t* y, int rows, int d, float eps, const int* row_map, hipStream_t s, int dt = DT_BF16,
                    const QuantActArgs* qa = nullptr);
void launch_add_rmsnorm(bf16_t* x, const float* P, int ksplit, int mpad, const float* w, bf16_t* y, int rows, int d, float eps, hipStream_t s,
                        int dt = DT_BF16, const DeqInfo* dq = nullptr, const QuantOut* qo = nullptr, const PrefetchRange* pf = nullptr, int pf_blocks = 0);
void launch_rmsnorm_ss(const bf16_t* x, const float* SS, const float* w, bf16_t* y, int rows, int d, float eps, hipStream_t s, int dt = DT_BF16);
void launch_swiglu_slab(const float* P, int ksplit, int mpad, int n2, bf16_t* act, int rows, hipStream_t s, int dt = DT_BF16, int gu8 = 0);
// int8 decode: int32 gate/up slabs (rows interleaved in 16-row groups as for EPI_SWIGLU) -> act (fp16) + its quantised form
void launch_swiglu_quant(const float* P, int ksplit, int mpad, int ff, bf16_t* act, int rows, const DeqInfo& dq, const QuantOut& qo, hipStream_t s);
void launch_rope_enc(bf16_t* qk, long ld, int M, int T, int heads2, int hd, int rd, const float* cs, hipStream_t s, int dt = DT_BF16);
void launch_assemble_embeds(const int* src, const bf16_t* table, const bf16_t* audio, bf16_t* x, int n_tok, int d, hipStream_t s);
void launch_fill_i32(int* p, int value, int n, hipStream_t s);

// device-resident ingest (ingest.hip): stage windows from per-session rings of raw wire PCM with the reference's a1 + a2 arithmetic
#define RING_MAX_WIN 64
struct RingStageArgs {
    const short* ring[RING_MAX_WIN];   // per window: ring base, or null for a host window (already staged by the caller)
    long ring_cap[RING_MAX_WIN];       // ring capacity in samples
    long start[RING_MAX_WIN];          // position of the window's first sample in the ring (absolute index modulo capacity)
    int n[RING_MAX_WIN];               // samples in the window
    int req_of[RING_MAX_WIN];          // request the window belongs to (the peak is per request)
    int* peak;                         // [requests] max |s|, zeroed by the caller
    short* pcm; long win_cap;          // engine PCM staging [W][win_cap]
};
void launch_ring_stage(const RingStageArgs& a, int W, int max_n, hipStream_t s);
void launch_f32_to_bf16(const float* in, bf16_t* out, long n, hipStream_t s, int dt = DT_BF16);      // fp32 -> element type
void launch_bf16_to_f32(const bf16_t* in, float* out, long n, hipStream_t s, int dt = DT_BF16);      // element type -> fp32
void launch_bf16_to_f16(const bf16_t* in, bf16_t* out, long n, hipStream_t s);                       // bf16 storage -> fp16 storage (RNE), in place allowed
void launch_synth_fill(unsigned long long key, long n, float scale, float offset, bf16_t* out_bf, float* out_f32, hipStream_t s, int round_f32 = 0);   // round_f32: the fp32 output holds the bf16-rounded value

// ---- int8 mode: activation / weight quantisation (quant.hip) ----
// Row-wise int8 of a weight matrix [N][K] (fp16 storage): CB, SCB (Int8Params.cuda(): int8_vectorwise_quant(W.half()))
void launch_quant_weights(const bf16_t* w, int8_t* cb, float* scb, int N, int K, hipStream_t s);
// One Linear8bitLt input X [M][K] (fp16 storage, row stride ld): outlier columns per GROUP of rows (= one reference call), row absmax
// without the outliers, int8 rows.  flags: scratch [G][K] bytes.  group of row r = gmap ? gmap[r / gdiv] : r / gdiv.
struct QuantActArgs {
    const bf16_t* X; long ld; int M, K;
    const int* gmap; int gdiv; int G;
    unsigned char* flags;
    int8_t* q; float* sca; int* oc_cnt; int* oc_list; int oc_ld;
};
void launch_quant_act(const QuantActArgs& a, hipStream_t s);
// The same result in two halves when the producer of X owns whole rows (LayerNorm): `begin` clears the group flags, the producer writes X and, in
// the same pass, the row's absmax, its int8 codes (its own elements >= 6.0 as 0) and the flags of those elements (launch_layernorm's `qa`);
// `finish` lists the groups' outlier columns and zeroes them in the rows that held smaller values there.  Two streaming passes over X fewer.
void launch_quant_act_begin(const QuantActArgs& a, hipStream_t s);
void launch_quant_act_finish(const QuantActArgs& a, hipStream_t s);
// finishes the rows an int8 GEMM deferred (GemmI8::defer_out): dense fp16 MFMA product over their gathered outlier columns + residual
void launch_i8_outlier_side(const GemmArgs& g, hipStream_t s);
// decode flavour: every row is its own group; one block per row
void launch_quant_rows(const bf16_t* X, long ld, int M, int K, const QuantOut& qo, hipStream_t s);
void launch_tile_weights_i8(const int8_t* w, int8_t* wt, int N, int K, hipStream_t s);
void launch_transpose_i8(const int8_t* w, int8_t* wt, int N, int K, hipStream_t s);     // wt[k][n] = w[n][k]
// int8 encoder: V columns [col0, col0 + C) of the row-major QKV matrix -> V^T [seg][C][vt_ld]
void launch_transpose_v(const bf16_t* qkv, long ld, int col0, bf16_t* vt, int n_seg, int T, int C, int vt_ld, long vt_seg_stride, hipStream_t s);

// ---- SONIC_MODE_F32 (test only; f32kind.hip): plain fp32 stages behind the same C ABI ----
enum { F32_EPI_NONE = 0, F32_EPI_GELU = 1, F32_EPI_RESID = 2 };
struct F32Gemm {
    const float* A; long lda, sA1, sA2;        // A[b][m][k]: rows K-contiguous, row stride lda (overlapping rows allowed), batch offsets per level
    const float* W; long swn, swk, sW1, sW2;   // W[b][n * swn + k * swk]; swn = swk = 0 means torch Linear layout (swn = K, swk = 1)
    float* C; long ldc, sC1, sC2;
    const float* bias;                         // [N] or null
    const float* R; long ldr;                  // F32_EPI_RESID: C = R + (acc + bias)
    int M, N, K, nb1, nb2;                     // batch = nb1 * nb2 (0 = 1)
    float scale;                               // acc * scale before the bias (0 = 1)
    int epi;
};
void launch_f32_gemm(const F32Gemm& g, hipStream_t s);
struct F32Attn {
    const float* Q; long ldq;                  // [n_tok][Hq * hd]
    const float* K; const float* V; long ldkv, seq_stride;   // [seq][key][Hkv * hd]
    float* O; long ldo;
    const int* seq; int seq_div;               // sequence of token t: seq ? seq[t] : t / seq_div
    const int* pos; int lim_const, lim_max;    // keys visible to token t: pos ? pos[t] + 1 : lim_const (both capped at lim_max, which sizes the LDS)
    int hd, grp;                               // head dim (<= 256, divides 256), query heads per kv head
    float scale;
};
void launch_f32_attn(const F32Attn& a, int n_tok, int heads, hipStream_t s);
void launch_f32_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int d, float eps, hipStream_t s);
void launch_f32_rmsnorm(const float* x, const float* w, float* y, int rows, int d, float eps, const int* row_map, hipStream_t s);
void launch_f32_rope(float* x, long ld, int n_tok, int heads, int hd, int rd, const float* cs, const int* pos, int pos_mod, hipStream_t s);
void launch_f32_feats_tm(const float* in, float* out, int W, int n_mels, int n_frames, hipStream_t s);
void launch_f32_conv_w(const float* in, float* out, int C, int Ci, hipStream_t s);
void launch_f32_zero_pad_rows(float* h, int W, int n_frames, int C, hipStream_t s);
void launch_f32_assemble(const int* src, const float* table, const float* audio, float* x, int n_tok, int d, hipStream_t s);
void launch_f32_kv_append(const float* kn, const float* vn, float* Kc, float* Vc, const int* seq, const int* pos, int n_tok, int kd, long seq_stride, hipStream_t s);
void launch_f32_swiglu(const float* g, const float* u, float* act, long n, hipStream_t s);

// ---- the 1 - 4 row token step as weight-streaming GEMV kernels (gemv.hip; option decode_gemv) ----
#define GEMV_MAX_ROWS 4
enum { GEMV_SLAB_NORM = 0 /* P[r][n] = (RMSNorm(X) . norm_w) W^T, fp32 */, GEMV_RESID = 1 /* resid += X W^T */, GEMV_SWIGLU_NORM = 2 /* act = silu(g) * u of (RMSNorm(X) . norm_w) Wgu^T */ };
struct GemvArgs {
    const bf16_t* X; long ldx;       // [M <= 4][K] activation rows (the raw residual stream for the *_NORM modes)
    const bf16_t* W;                 // [N][K] fragment-tiled (launch_tile_weights; gate/up: launch_tile_weights_gu8)
    int M, N, K, dt;
    const float* norm_w; float eps;  // *_NORM modes: RMSNorm weight [K]
    float* P;                        // GEMV_SLAB_NORM: fp32 [M][N] (one "slab" for decode_attn_kernel / greedy_kernel)
    bf16_t* resid; long ldr;         // GEMV_RESID: residual rows, updated in place
    bf16_t* act;                     // GEMV_SWIGLU_NORM: [M][N / 2]
};
void launch_gemv(const GemvArgs& a, int mode, hipStream_t s);
bool gemv_eligible(int M, int N, int K, int mode);
