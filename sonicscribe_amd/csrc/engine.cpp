// sonic_hip engine: weights, HBM buffers, the mel -> encoder -> projector -> prefill -> greedy-decode
// pipeline on one HIP stream, hipGraph-captured decode step, and the C ABI of include/sonic_hip.h.
// One engine = one full model replica on one MI355X (replicas, no collectives).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sonic_hip.h"
#include "common.h"
#include "kernels.h"

#define T_PAD_ALIGN 64
#define CHK_RING 64            // check events / pinned n_active words of the decode loop
#define DEV_ERR_ACTIVE (-(1 << 23))   // a running-row count below this: the device error word was set (greedy_kernel), the step's outputs are invalid
#define SVC_WORDS 132           // per check of the continuous loop: finished[64], n_new[64], n_active, padding
#define CHK_MAX_AHEAD 32       // deepest lookahead in chunks (a host that is frozen for tens of ms at a time - CPU quota, a busy event loop)

static thread_local std::string g_create_err;

struct DevTensor {
    std::vector<int64_t> shape;
    bf16_t* p = nullptr;
    size_t n = 0;
};

// int8 mode (asr.py:169-210): a quantised Linear keeps row-wise int8 weights + row absmax instead of its 16-bit matrix
struct QW { int8_t* cb = nullptr; float* scb = nullptr; int8_t* cbt = nullptr; int8_t* cbk = nullptr; bool cb_rowmajor_kept = true; };   // cbt: fragment-tiled copy, cbk: k-major copy (decode step)
struct EncLayerW { float *ln1w, *ln1b, *bqkv, *bo, *ln2w, *ln2b, *b1, *b2; bf16_t *wqkv, *wo, *w1, *w2; QW qqkv, qo, q1, q2; };
#define KT_SLOT_BLOCKS 512                                   // "ktrace" diagnostics: blocks recorded per kernel slot, 8 timestamps each
struct DecLayerW { float *ln1, *ln2; bf16_t *wqkv, *wo, *wgu, *wdown;          // row-major (prefill GEMM)
                   bf16_t *wqkv_t, *wo_t, *wgu_t, *wgu_t8, *wdown_t;                    // fragment-tiled copies (decode skinny GEMM)
                   QW qqkv, qo, qgu, qdown; };

// SONIC_MODE_F32 (test only, f32kind.hip): fp32 weights as loaded (torch Linear layout, nothing packed but the conv taps) and fp32 activation buffers
struct F32EncL { float *ln1w, *ln1b, *wq, *bq, *wk, *wv, *bv, *wo, *bo, *ln2w, *ln2b, *w1, *b1, *w2, *b2; };
struct F32DecL { float *ln1, *wq, *wk, *wv, *wo, *ln2, *wg, *wu, *wd; };
struct F32State {
    std::map<std::string, float*> raw;               // name -> device tensor (owned by the engine's alloc list)
    float *conv1w = nullptr, *conv1b = nullptr, *conv2w = nullptr, *conv2b = nullptr, *enc_nw = nullptr, *enc_nb = nullptr;
    float *pj1w = nullptr, *pj1b = nullptr, *pj2w = nullptr, *pj2b = nullptr, *embed = nullptr, *dec_nw = nullptr;
    std::vector<F32EncL> enc; std::vector<F32DecL> dec;
    // encoder: time-major padded features, conv1 output (padded), residual stream, norm output, q / k / v, attention output, MLP, projector
    float *featT = nullptr, *h1 = nullptr, *x = nullptr, *ln = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *att = nullptr, *ff = nullptr, *ph = nullptr, *pe = nullptr;
    // decoder: token rows of the prefill (or the R rows of a token step), KV cache [layer][seq][ctx][KD], logits [64][vocab]
    float *dx = nullptr, *dhn = nullptr, *dq = nullptr, *dk = nullptr, *dv = nullptr, *datt = nullptr, *dg = nullptr, *du = nullptr, *dact = nullptr;
    float *Kc = nullptr, *Vc = nullptr, *logits = nullptr, *hlast = nullptr;
};

struct sonic_engine {
    sonic_dims d;
    int device = 0, mode = 0, Bm = 0, max_ctx = 0;
    int dt = DT_BF16;          // activation / 16-bit weight element type: bf16 (native) or fp16 (int8 mode, asr.py:61)
    bool i8 = false;           // LLM.int8 linears
    bool f32 = false;          // SONIC_MODE_F32: the fp32 kind of every stage (test only; F32State, f32kind.hip)
    F32State* f = nullptr;
    hipStream_t st = nullptr;
    std::mutex mu;
    std::string err;
    std::vector<void*> allocs;
    bool cap_svc = false;                           // a continuous loop's chunk graph is being captured (chunk_graph)
    int64_t weight_bytes = 0, alloc_bytes = 0;      // alloc_bytes: every live device allocation of this engine (sonic_memory_info)
    bool finalized = false;

    std::map<std::string, DevTensor> raw;
    // packed weights
    bf16_t *conv1w = nullptr, *conv2w = nullptr; float *conv1b = nullptr, *conv2b = nullptr;
    std::vector<EncLayerW> enc;
    float *enc_nw = nullptr, *enc_nb = nullptr;
    unsigned short* gelu_lut = nullptr;     // bf16 GELU table (gemm256 EPI_BIAS_GELU epilogue), native mode
    bf16_t *pj1w = nullptr, *pj2w = nullptr; float *pj1b = nullptr, *pj2b = nullptr; QW qpj1, qpj2;
    // int8 mode scratch: quantised activations of the GEMM in flight, row statistics, outlier columns per request, window -> request map
    int8_t* qa = nullptr; float* q_sca = nullptr; unsigned char* q_flags = nullptr; int *q_oc_cnt = nullptr, *q_oc_list = nullptr, *win_req = nullptr;
    int q_kmax = 0; bf16_t* qkv_rm = nullptr;
    bf16_t* defer_tmp = nullptr; size_t defer_cap = 0; int opt_i8_defer_thr = 8; int opt_i8_no_xq = 0; int opt_i8_no_lnq = 0; int opt_i8_dbg = 0; int opt_i8_no_qkv_fuse = 0;   // rows deferred to the outlier side product leave the GEMM here
    // int8 decode step: the three quantised row sets (input norm output, attention output, SwiGLU output)
    int8_t *hn_q = nullptr, *att_q = nullptr, *act_q = nullptr; float *sca_hn = nullptr, *sca_att = nullptr, *sca_act = nullptr;
    int *oc_hn = nullptr, *oc_att = nullptr, *oc_act = nullptr, *ol_hn = nullptr, *ol_att = nullptr, *ol_act = nullptr;
    float *ov_hn = nullptr, *ov_att = nullptr, *ov_act = nullptr;      // the outliers' values beside the lists
    int* big_att = nullptr;                                             // [64][4] per-block counts of attention outputs >= 6.0
    float *amax_att = nullptr, *amax_act = nullptr;                     // [64][4] partial row maxima written by producers that do not own whole rows
    std::map<std::string, bool> raw_f16;   // int8 mode: tensors already converted to fp16 at load
    bf16_t* embed = nullptr; bf16_t* embed_t = nullptr;
    std::vector<DecLayerW> dec;
    float* dec_nw = nullptr;

    // derived sizes
    int T = 0, Tp = 0, Ta = 0, hd_e = 0, qkvN = 0, QD = 0, KD = 0, tok_cap = 0, out_cap = 0;
    long slabN = 0;

    // front-end
    int16_t* pcm = nullptr; int* n_samples_d = nullptr; std::vector<int> n_samples_h; int W = 0;
    float* logspec = nullptr; int* segmax = nullptr; bf16_t* feats_fm = nullptr; float* feats_f32 = nullptr;
    LogmelConst lc{};
    // encoder
    bf16_t *h1 = nullptr, *x = nullptr, *ln = nullptr, *qk = nullptr, *vt = nullptr, *att = nullptr, *ff = nullptr, *ph = nullptr, *pe = nullptr;
    float* enc_cs = nullptr;
    // decoder
    bf16_t *dx = nullptr, *dhn = nullptr, *dqkv = nullptr, *dq = nullptr, *datt = nullptr, *dact = nullptr;
    bf16_t *Kc = nullptr, *Vc = nullptr, *Vts = nullptr;
    float* dec_cs = nullptr;
    float *slab = nullptr, *lslab = nullptr, *ssq = nullptr;
    float* slab2 = nullptr;                                         // down_proj's slabs when the next q|k|v (or the lm_head) consumes them itself (PRE form, <= 2 rows)
    bf16_t *sx = nullptr, *shn = nullptr, *sq = nullptr, *satt = nullptr, *sact = nullptr;
    bf16_t* sx2 = nullptr;                                          // ... and the second residual buffer of that form (the stream ping-pongs layer by layer)
    int *kv_len = nullptr, *tok_pos = nullptr, *n_new = nullptr, *finished = nullptr, *max_new_d = nullptr, *n_active = nullptr;
    int *out_ids = nullptr, *step_ctr = nullptr, *seq_iota = nullptr;
    int *src = nullptr, *tok_seq = nullptr, *tok_pos_pf = nullptr, *q_off = nullptr, *q_len = nullptr, *last_row = nullptr;
    float* dump = nullptr; size_t dump_cap = 0; int dump_steps = 0;
    bf16_t* taps = nullptr; int taps_on = 0; int last_ntok = 0;   // debug: prefill hidden states after embedding + each layer
    int* n_active_h = nullptr;  // pinned
    int R = 0, max_steps = 0, greedy_calls = 0, steps_run = 0; bool run_logits = false;   // state of the staged batch between the stage entry points
    int* force_d = nullptr; int force_ld = 0, force_R = 0;   // teacher forcing for the next runs (sonic_set_forced_ids)
    std::vector<int> last_qlen, last_maxnew;
    std::map<std::pair<int, int>, hipGraphExec_t> graphs;        // (rows, token steps) -> captured chunk of the decode loop
    // the pipelined early-stop check: behind every chunk the device's count of running rows is copied to n_active_h[chunk % CHK_RING] and an
    // event is recorded; the host reads check k only when chunk k + lookahead is already queued (run_decode_steps)
    hipEvent_t chk_ev[CHK_RING]{};
    bool run_starved = false;                                      // the current batch saw the queue run dry (lookahead grew)
    int lookahead = 1;                                             // chunks queued beyond the one whose check the host waits for; adapts (1..CHK_MAX_AHEAD)
    int* plan_h = nullptr; size_t plan_cap = 0;                    // pinned staging of a batch's prompt plan (no stream synchronise between encoder and prefill)
    // Two staging buffers used alternately, each with an event recorded behind the last copy that reads it: sonic_prefill_enqueue returns with those
    // copies still queued behind the encoder, and the next run on this handle must not overwrite a buffer the stream has not read yet
    int* plan_buf[2] = {nullptr, nullptr}; hipEvent_t plan_ev[2] = {nullptr, nullptr}; bool plan_busy[2] = {false, false}; int plan_idx = 0;

    // Slots (sonic_slot_create): further in-flight batches on ONE weight copy.  A slot is an engine of its own in every respect - stream, activation
    // buffers, KV cache, PCM staging, decode graphs, lock, options - except that its weight / constant pointers are the owner's.
    sonic_engine* owner = nullptr;                                 // slot: whose weights these are (never a slot itself)
    std::vector<sonic_engine*> slots;                              // owner: its slots (destroyed with it at the latest)
    std::mutex rings_mu;                                           // owner: guards `rings` (the registry every slot stages from)

    // continuous decoding (sonic_service_*): this engine's rows are a pool - requests prefilled on another handle of the same weights are spliced
    // into free rows between chunks of an endless greedy loop, finished rows are fetched and freed one by one
    bool svc_on = false;
    int64_t svc_launched = 0, svc_checked = 0;                     // chunks queued / checks read since sonic_service_begin
    int64_t svc_dry = 0;                                           // launches that found the stream empty with rows running (diagnostics)
    int svc_calm = 0;                                              // chunks since the queue last ran dry (the lookahead shrinks again after 256 of them)
    int* svc_h = nullptr;                                          // pinned ring [CHK_RING][SVC_WORDS]: finished[64] | n_new[64] | n_active
    int svc_fin[64]{}, svc_nn[64]{}, svc_active = 0; int64_t svc_seq = 0;   // the newest check read: state after chunk number svc_seq
    hipEvent_t sync_ev = nullptr;                                  // blocking-sync event behind stream_sync()
    hipStream_t st_io = nullptr;                                   // row fetches (a finished row's ids are stable: no ordering against the queued chunks needed)
    hipEvent_t xfer_ev = nullptr, splice_ev = nullptr, wait_ev = nullptr; bool wait_pending = false;   // cross-handle ordering of a splice

    // sonic_run_staged_async / sonic_wait: a worker thread of the engine's own runs the batch, the caller's thread returns at once
    struct AsyncJob { std::vector<int32_t> req_win, prompt_ids, max_new; std::vector<int64_t> prompt_off; int R = 0; bool has_rw = false; int want_logits = 0; } a_job;
    std::thread a_thread; std::mutex a_mu; std::condition_variable a_cv;
    bool a_started = false, a_pending = false, a_running = false, a_done = false, a_stop = false; int a_status = 0;

    // experiment knobs (sonic_set_option): per engine, copied into the launchers' thread-local view by ENTER()
    LaunchOpts opts;
    int opt_no_graph = 0, opt_gemm_timing = 0, opt_no_fused_rope = 0, opt_no_gelu_lut = 0, opt_gemm_trace = 0, opt_no_rope_tiles = 0, opt_prefill_rowmajor = 0;
    double host_launch_ms = 0, host_wait_ms = 0; int host_launches = 0;   // host time of the last run's decode loop: inside hipGraphLaunch / waiting for a check
    int step_launches_per_layer = 0;   // of the token step built last (decode_step): sonic_timings.decode_launches_per_layer
    int opt_f32_synth_bf16 = 0;    // SONIC_MODE_F32: sonic_load_synthetic writes the bf16-rounded values (the weights a bf16 engine gets from the same seed) as fp32
    int opt_decode_gemv = 0;       // 1: token steps of <= 4 rows run the GEMV chain (gemv.hip: five launches per layer, no MFMA tiles); opt-in - another summation order than the
                                   // MFMA chain, so a request's low bits then depend on whether its step had <= 4 rows.  Measured slower than the MFMA chain (profiles/round6_gemv_ab.txt): an experiment, off
    int opt_no_pre_norm = 0;       // 1: never the PRE form of the <= 2-row decode step (standalone add+RMSNorm launches as for more rows; A/B - same bits)
    int opt_decode_chunk = 2;      // token steps per captured graph = granularity of the early-stop check (sonic_set_option "decode_chunk")
    long long* kt = nullptr; int kt_layer = -1;     // diagnostics ("ktrace" option): in-kernel timestamps of one decoder layer's kernels
    int* ring_peak = nullptr;                        // [Bm] per-request max |s| of a ring-staged batch (ingest.hip)
    std::vector<struct sonic_ring*> rings;          // rings created on this engine and not yet destroyed (freed with the engine at the latest)

    // timing
    hipEvent_t ev[5]{};
    std::vector<hipEvent_t> gemm_ev;
    int gemm_ev_used = 0;
    sonic_timings tim{};
};

// ------------------------------------------------------------------------------------------ helpers
static int fail(sonic_engine* e, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (e) e->err = buf; else g_create_err = buf;
    return code;
}
#define HIPC(e, call)                                                                                      \
    do {                                                                                                   \
        hipError_t _r = (call);                                                                            \
        if (_r != hipSuccess) {                                                                            \
            const bool oom = (_r == hipErrorOutOfMemory);                                                  \
            return fail(e, oom ? SONIC_ERR_OOM : SONIC_ERR_HIP, "%s%s failed: %s (%s:%d)",                 \
                        oom ? "HIP out of memory: " : "", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
        }                                                                                                  \
    } while (0)

// Wait for the engine stream WITHOUT spinning.  hipStreamSynchronize busy-waits on this runtime when the machine shows more CPUs than contexts
// (256 visible): every waiting thread of every slot / rank burnt a CPU - 3.7 CPUs for three slots of one rank, far beyond a 16-CPU quota at eight
// ranks, and a job that exhausts its quota has ALL its threads frozen.  An event created with hipEventBlockingSync sleeps on an
// interrupt instead; the wake-up is slower by tens of microseconds, which the queued work hides (the decode loop keeps `lookahead` chunks ahead).
static hipError_t stream_sync(sonic_engine* e);

// Host -> device copy on the ENGINE stream, complete on return.  Never use the null-stream hipMemcpy for uploads: the
// engine stream is non-blocking, so a null-stream copy is not ordered against work still queued on it (dalloc's zero fill
// once wiped parts of a freshly uploaded RoPE table that way).
static hipError_t h2d(sonic_engine* e, void* dst, const void* src, size_t bytes) {
    hipError_t r = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->st);
    if (r != hipSuccess) return r;
    return stream_sync(e);
}

// zero fill by a kernel on the engine stream (hipMemsetAsync on a non-blocking stream: see TmpBuf::get)
// Device -> host on the ENGINE stream.  The synchronous hipMemcpy goes through the legacy stream, which implicitly synchronises with
// other streams of the process: with several engines in one process (replicas on one or several GPUs) it failed with "operation would
// make the legacy stream depend on a capturing blocking stream" while another engine's thread was capturing its decode graph.
static hipError_t d2h_async(sonic_engine* e, void* dst, const void* src, size_t bytes) {
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->st);
}
static hipError_t d2h(sonic_engine* e, void* dst, const void* src, size_t bytes) {
    hipError_t r = d2h_async(e, dst, src, bytes);
    return r == hipSuccess ? stream_sync(e) : r;
}
static hipError_t stream_sync(sonic_engine* e) {
    if (!e->sync_ev) return hipStreamSynchronize(e->st);
    hipError_t r = hipEventRecord(e->sync_ev, e->st);
    return r == hipSuccess ? hipEventSynchronize(e->sync_ev) : r;
}
static void zero_fill(sonic_engine* e, void* q, size_t bytes) {
    size_t left = bytes / 4; int* w = (int*)q;
    while (left > 0) { const int c = left > (1u << 30) ? (1 << 30) : (int)left; launch_fill_i32(w, 0, c, e->st); w += c; left -= c; }
}
template <typename Tt> static int dalloc(sonic_engine* e, Tt** p, size_t n, bool zero = true) {
    void* q = nullptr;
    const size_t bytes = ((n ? n : 1) * sizeof(Tt) + 3) / 4 * 4;
    HIPC(e, hipMalloc(&q, bytes));
    e->allocs.push_back(q); e->alloc_bytes += (int64_t)bytes;
    if (zero) zero_fill(e, q, bytes);
    *p = (Tt*)q;
    return SONIC_OK;
}
// Every device buffer is an ordinary hipMalloc allocation that sonic_destroy hands back to the driver (`del asr_model.model` is expected to
// return the VRAM: backend/main.py:84-90).  Rounds 2-3 kept the per-step activation buffers (and, in round 2, the KV cache and the tiled
// weights) in uncached (MTYPE UC) memory, parked in a process-wide pool that was never freed, because memory recycled between uncached and
// ordinary allocations came back with stale cache lines on this stack (tests/test_gemm256_path failed in 3 of 5 full-suite runs; root cause
// never found, tools/uc_recycle_repro.hip excludes two mechanisms).  Round 4 measured what the uncached buffers were still worth: nothing
// (bench.py, one box, SONIC_NO_UC=1 vs default: 126.27 vs 126.17 segments/s with two slots, 103.3 vs 103.6 one batch at a time - inside the
// run-to-run spread), so the uncached path, its pool and sonic_release_pool are gone rather than kept alive for an unmeasurable gain.
template <typename Tt> static int dalloc_act(sonic_engine* e, Tt** p, size_t n) { return dalloc(e, p, n, true); }
template <typename Tt> static int dalloc_big(sonic_engine* e, Tt** p, size_t n, bool zero = true) { return dalloc(e, p, n, zero); }
#define TRY(x) do { int _s = (x); if (_s != SONIC_OK) return _s; } while (0)
// every locked C-ABI entry: serialise on the engine, select its device, and hand its experiment knobs to the launchers
// (hipGetLastError first: the slot is per thread and sticky, so a failure some earlier call of this thread ignored would otherwise
// surface at this call's closing hipGetLastError check)
#define ENTER(e) std::lock_guard<std::mutex> lk((e)->mu); (void)hipGetLastError(); HIPC(e, hipSetDevice((e)->device)); g_opts = (e)->opts

static inline float bf16_round_host(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    memcpy(&x, &u, 4); return x;
}
static uint64_t mix64h(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
static uint64_t fnv1a64h(const char* s) { uint64_t h = 0xCBF29CE484222325ULL; for (; *s; ++s) { h ^= (uint8_t)*s; h *= 0x100000001B3ULL; } return h; }

// small in-file kernels ---------------------------------------------------------------------------
// [B][n_mels][n_frames] fp32 (HF layout) -> frame-major bf16 with one zero row each side
template <typename T> __global__ void feats_to_fm_kernel(const float* in, T* out, int n_mels, int n_frames) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)n_mels * n_frames) return;
    const int t = e / n_mels, m = e % n_mels;
    out[((long)b * (n_frames + 2) + 1 + t) * n_mels + m] = (T)in[((long)b * n_mels + m) * n_frames + t];   // asr.py:280-301: cast to the model dtype
}
// conv weight [C][Ci][3] -> [C][3][Ci]
__global__ void conv_permute_kernel(const bf16_t* in, bf16_t* out, int C, int Ci) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * Ci * 3) return;
    const int k = e % 3, ci = (e / 3) % Ci, c = e / (3L * Ci);
    out[((long)c * 3 + k) * Ci + ci] = in[e];
}

// ------------------------------------------------------------------------------------------ tensor inventory (spec.py order)
struct InvEntry { std::string name; std::vector<int64_t> shape; int kind; };  // kind 0 mat, 1 embed, 2 bias, 3 norm
static std::vector<InvEntry> inventory(const sonic_dims& d) {
    std::vector<InvEntry> v;
    const int64_t C = d.enc_d, F = d.enc_ff, M = d.n_mels;
    auto add = [&](const std::string& n, std::vector<int64_t> s, int k) { v.push_back({n, s, k}); };
    const std::string at = "model.audio_tower.";
    add(at + "conv1.weight", {C, M, 3}, 0); add(at + "conv1.bias", {C}, 2);
    add(at + "conv2.weight", {C, C, 3}, 0); add(at + "conv2.bias", {C}, 2);
    for (int i = 0; i < d.enc_layers; ++i) {
        const std::string p = at + "layers." + std::to_string(i) + ".";
        add(p + "input_layernorm.weight", {C}, 3); add(p + "input_layernorm.bias", {C}, 2);
        add(p + "self_attn.q_proj.weight", {C, C}, 0); add(p + "self_attn.q_proj.bias", {C}, 2);
        add(p + "self_attn.k_proj.weight", {C, C}, 0);
        add(p + "self_attn.v_proj.weight", {C, C}, 0); add(p + "self_attn.v_proj.bias", {C}, 2);
        add(p + "self_attn.o_proj.weight", {C, C}, 0); add(p + "self_attn.o_proj.bias", {C}, 2);
        add(p + "post_attention_layernorm.weight", {C}, 3); add(p + "post_attention_layernorm.bias", {C}, 2);
        add(p + "mlp.fc1.weight", {F, C}, 0); add(p + "mlp.fc1.bias", {F}, 2);
        add(p + "mlp.fc2.weight", {C, F}, 0); add(p + "mlp.fc2.bias", {C}, 2);
    }
    add(at + "norm.weight", {C}, 3); add(at + "norm.bias", {C}, 2);
    const int64_t PI = C * d.merge, PM = 2L * d.dec_d, D = d.dec_d;
    const std::string pj = "model.multi_modal_projector.";
    add(pj + "linear_1.weight", {PM, PI}, 0); add(pj + "linear_1.bias", {PM}, 2);
    add(pj + "linear_2.weight", {D, PM}, 0); add(pj + "linear_2.bias", {D}, 2);
    const std::string lm = "model.language_model.";
    add(lm + "embed_tokens.weight", {d.vocab, D}, 1);
    const int64_t QD = (int64_t)d.dec_heads * d.dec_head_dim, KD = (int64_t)d.dec_kv_heads * d.dec_head_dim, FF = d.dec_ff;
    for (int i = 0; i < d.dec_layers; ++i) {
        const std::string p = lm + "layers." + std::to_string(i) + ".";
        add(p + "input_layernorm.weight", {D}, 3);
        add(p + "self_attn.q_proj.weight", {QD, D}, 0); add(p + "self_attn.k_proj.weight", {KD, D}, 0);
        add(p + "self_attn.v_proj.weight", {KD, D}, 0); add(p + "self_attn.o_proj.weight", {D, QD}, 0);
        add(p + "post_attention_layernorm.weight", {D}, 3);
        add(p + "mlp.gate_proj.weight", {FF, D}, 0); add(p + "mlp.up_proj.weight", {FF, D}, 0); add(p + "mlp.down_proj.weight", {D, FF}, 0);
    }
    add(lm + "norm.weight", {D}, 3);
    return v;
}
static size_t numel(const std::vector<int64_t>& s) { size_t n = 1; for (auto x : s) n *= (size_t)x; return n; }

// ------------------------------------------------------------------------------------------ constants
static int build_constants(sonic_engine* e) {
    const sonic_dims& d = e->d;
    std::vector<float> win(400), ct(400), stb(400);
    for (int i = 0; i < 400; ++i) {
        win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 400.0));   // torch.hann_window(400), periodic
        ct[i] = (float)cos(2.0 * M_PI * i / 400.0);
        stb[i] = (float)sin(2.0 * M_PI * i / 400.0);
    }
    // slaney mel bank, HF:audio_utils.py:448-518,541-560,690-722 -> CSR per filter
    const int nm = d.n_mels, nf = nm + 2;
    auto hz2mel = [](double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; };
    auto mel2hz = [](double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; };
    std::vector<double> ffq(nf);
    const double mmin = hz2mel(0.0), mmax = hz2mel(8000.0);
    for (int i = 0; i < nf; ++i) ffq[i] = mel2hz(i == nf - 1 ? mmax : mmin + (mmax - mmin) / (nf - 1) * i);
    std::vector<int> lo(nm), cnt(nm), off(nm);
    std::vector<float> w;
    for (int m = 0; m < nm; ++m) {
        int first = -1, last = -2;
        std::vector<float> taps(201);
        for (int b = 0; b < 201; ++b) {
            const double fb = (b == 200) ? 8000.0 : 8000.0 / 200.0 * b;
            const double down = -(ffq[m] - fb) / (ffq[m + 1] - ffq[m]), up = (ffq[m + 2] - fb) / (ffq[m + 2] - ffq[m + 1]);
            double v = down < up ? down : up; if (v < 0) v = 0;
            v *= 2.0 / (ffq[m + 2] - ffq[m]);
            taps[b] = (float)v;
            if (taps[b] != 0.f) { if (first < 0) first = b; last = b; }
        }
        if (first < 0) { first = 0; last = -1; }
        lo[m] = first; cnt[m] = last - first + 1; off[m] = (int)w.size();
        for (int b = first; b <= last; ++b) w.push_back(taps[b]);
    }
    float *dwin, *dct, *dst, *dw; int *dlo, *dcnt, *doff;
    TRY(dalloc(e, &dwin, 400)); TRY(dalloc(e, &dct, 400)); TRY(dalloc(e, &dst, 400)); TRY(dalloc(e, &dw, w.size()));
    TRY(dalloc(e, &dlo, nm)); TRY(dalloc(e, &dcnt, nm)); TRY(dalloc(e, &doff, nm));
    HIPC(e, hipMemcpyAsync(dwin, win.data(), 1600, hipMemcpyHostToDevice, e->st));
    HIPC(e, hipMemcpyAsync(dct, ct.data(), 1600, hipMemcpyHostToDevice, e->st));
    HIPC(e, hipMemcpyAsync(dst, stb.data(), 1600, hipMemcpyHostToDevice, e->st));
    HIPC(e, hipMemcpyAsync(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice, e->st));
    HIPC(e, hipMemcpyAsync(dlo, lo.data(), nm * 4, hipMemcpyHostToDevice, e->st));
    HIPC(e, hipMemcpyAsync(dcnt, cnt.data(), nm * 4, hipMemcpyHostToDevice, e->st));
    HIPC(e, hipMemcpyAsync(doff, off.data(), nm * 4, hipMemcpyHostToDevice, e->st));
    HIPC(e, stream_sync(e));
    e->lc = LogmelConst{dwin, dct, dst, dlo, dcnt, doff, dw};

    // RoPE tables: cos/sin computed in fp32 and cast to the activation dtype (modeling_glmasr.py:95-106)
    auto round_act = [&](float x) -> float { return e->f32 ? x : e->dt == DT_F16 ? (float)(_Float16)x : bf16_round_host(x); };
    auto rope_table = [&](int n_pos, int rd, float theta, float** out) -> int {
        const int half = rd / 2;
        std::vector<float> t((size_t)n_pos * rd);
        for (int p = 0; p < n_pos; ++p)
            for (int i = 0; i < half; ++i) {
                const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)rd);
                const float ang = inv * (float)p;
                t[(size_t)p * rd + i] = round_act(cosf(ang));          // `cos.to(dtype=x.dtype)`: bf16 (native), fp16 (int8 / fp16 modes)
                t[(size_t)p * rd + half + i] = round_act(sinf(ang));
            }
        TRY(dalloc(e, out, t.size()));
        HIPC(e, h2d(e, *out, t.data(), t.size() * 4));
        return SONIC_OK;
    };
    if (e->dt == DT_BF16) {
        // GELU of every bf16 value with |x| in [2^-14, 16), computed exactly as the reference op sequence does it in fp32
        // (0.5 * x * (1 + erff(x / sqrt 2)), modeling_glmasr.py:299-300 / torch GELU(approximate="none")) and rounded to bf16 once
        std::vector<unsigned short> lut(GELU_LUT_N);
        for (int sgn = 0; sgn < 2; ++sgn)
            for (int i = 0; i < GELU_LUT_HALF; ++i) {
                const uint32_t bits = ((uint32_t)sgn << 31) | ((uint32_t)((GELU_LUT_E0 << 7) + i) << 16);
                float x; memcpy(&x, &bits, 4);
                const float y = bf16_round_host(0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)));
                uint32_t yb; memcpy(&yb, &y, 4);
                lut[sgn * GELU_LUT_HALF + i] = (unsigned short)(yb >> 16);
            }
        TRY(dalloc(e, &e->gelu_lut, GELU_LUT_N));
        HIPC(e, h2d(e, e->gelu_lut, lut.data(), GELU_LUT_N * 2));
    }
    TRY(rope_table(e->T, d.enc_rotary_dim, d.enc_theta, &e->enc_cs));
    TRY(rope_table(e->max_ctx, d.dec_head_dim, d.dec_theta, &e->dec_cs));
    return SONIC_OK;
}

// ------------------------------------------------------------------------------------------ create / destroy
// The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), round-robin in creation order; streams that
// share a queue execute in order.  An engine with two slots owns six streams (one main + one fetch stream per handle), plus one per device
// ring: with four queues the prefill slot's stream landed on the queue of a decoding handle and the bulk pipeline lost 7 % (141 vs 151
// segments/s, profiles/round4_hw_queues.txt).  The runtime reads the variable once, at its first call, so it is the HOST PROCESS that has to set GPU_MAX_HW_QUEUES=8
// before anything touches HIP (`import sonicscribe_amd` and bench.py do it with setdefault).  The library itself no longer writes the
// environment (round 4's constructor did: setenv from a library constructor races with other threads' getenv and silently did nothing when torch
// had started the runtime first).  Instead the first sonic_create on a device MEASURES how many hardware queues the process really has and says so
// once on stderr when there are fewer than the pipeline wants; sonic_runtime_info reports the same numbers to the host.
#define SONIC_HW_QUEUES_WANTED 8
__global__ void hwq_probe_kernel(unsigned long long ticks) {      // spins for `ticks` of the 100 MHz constant clock (s_memrealtime)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static std::mutex g_hwq_mu;
static int g_hwq_measured[64];                                      // per device: 0 = not probed yet
// Eight streams each get one single-wave kernel that spins 300 us; streams that share a hardware queue run in order, so the wall time of the eight
// is ceil(8 / queues) x 300 us.  Run once per device and process (about 1 ms), before the engine creates its own streams; the probe streams are
// created and destroyed in one go (a multiple of every plausible queue count, so the runtime's round-robin hand-out is where it was).
static int probe_hw_queues(int device) {
    std::lock_guard<std::mutex> lk(g_hwq_mu);
    if (device < 0 || device >= 64) return 0;
    if (g_hwq_measured[device]) return g_hwq_measured[device];
    int cur = 0; (void)hipGetDevice(&cur);
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    constexpr int NS = 8; constexpr unsigned long long TICKS = 30000;   // 300 us
    hipStream_t st[NS] = {};
    int made = 0, q = 0;
    for (; made < NS; ++made) if (hipStreamCreateWithFlags(&st[made], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
    if (made == NS) {
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {                           // rep 0 also loads the code object
            for (int i = 0; i < NS; ++i) (void)hipStreamSynchronize(st[i]);
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < NS; ++i) hipLaunchKernelGGL(hwq_probe_kernel, dim3(1), dim3(64), 0, st[i], rep == 0 ? 100ull : TICKS);
            for (int i = 0; i < NS; ++i) (void)hipStreamSynchronize(st[i]);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0 && us < best) best = us;
        }
        const double rounds = (best - 40.0) / (TICKS / 100.0);        // ~40 us of launch + synchronise overhead
        int r = (int)(rounds + 0.5); if (r < 1) r = 1; if (r > NS) r = NS;
        q = (NS + r - 1) / r;                                          // 1 round: >= 8 queues, 2 rounds: 4, 4 rounds: 2, 8 rounds: 1
        if (hipGetLastError() != hipSuccess) q = 0;
    }
    for (int i = 0; i < made; ++i) (void)hipStreamDestroy(st[i]);
    (void)hipSetDevice(cur);
    g_hwq_measured[device] = q;
    if (q > 0 && q < SONIC_HW_QUEUES_WANTED && !getenv("SONIC_QUIET")) {
        const char* env = getenv("GPU_MAX_HW_QUEUES");
        fprintf(stderr, "[sonic_hip] device %d: the HIP runtime of this process has %d hardware queue(s) for its streams (GPU_MAX_HW_QUEUES=%s%s); an engine with "
                        "slots wants %d - streams that share a queue run in order (the bulk pipeline measured 141 instead of 152 segments/s on 4 queues).  Set "
                        "GPU_MAX_HW_QUEUES=8 in the environment BEFORE the process first touches HIP / torch.cuda.\n", device, q, env ? env : "unset",
                env && atoi(env) >= SONIC_HW_QUEUES_WANTED ? ": set after the runtime had started" : "", SONIC_HW_QUEUES_WANTED);
    }
    return q;
}
extern "C" int sonic_runtime_info(int device_id, int32_t* hw_queues, int32_t* hw_queues_env, int32_t* hw_queues_wanted) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) { (void)hipGetLastError(); return fail(nullptr, SONIC_ERR_INVALID, "device %d not available", device_id); }
    if (hw_queues) *hw_queues = probe_hw_queues(device_id);
    if (hw_queues_env) { const char* v = getenv("GPU_MAX_HW_QUEUES"); *hw_queues_env = v ? atoi(v) : 0; }
    if (hw_queues_wanted) *hw_queues_wanted = SONIC_HW_QUEUES_WANTED;
    return SONIC_OK;
}

extern "C" int sonic_abi_version(void) { return SONIC_ABI_VERSION; }
extern "C" int sonic_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

static int check_dims(const sonic_dims& d, int max_batch, int max_ctx, int mode) {
    if (d.n_mels <= 0 || d.n_mels % 64 != 0) return fail(nullptr, SONIC_ERR_INVALID, "n_mels must be a positive multiple of 64");
    if (d.enc_d % 64 || d.enc_ff % 64 || d.dec_d % 256 || d.dec_ff % 256) return fail(nullptr, SONIC_ERR_INVALID, "hidden sizes must be multiples of 64 (encoder) / 256 (decoder)");
    if (d.enc_d / d.enc_heads != 64 || d.enc_d % d.enc_heads) return fail(nullptr, SONIC_ERR_INVALID, "encoder head_dim must be 64");
    if (d.dec_head_dim != 128) return fail(nullptr, SONIC_ERR_INVALID, "decoder head_dim must be 128");
    if (d.enc_rotary_dim % 16 || d.enc_rotary_dim > 64) return fail(nullptr, SONIC_ERR_INVALID, "encoder rotary dim must be a multiple of 16");
    if (d.dec_heads % d.dec_kv_heads || d.dec_heads / d.dec_kv_heads > 4) return fail(nullptr, SONIC_ERR_INVALID, "GQA group must divide and be <= 4");
    if (d.vocab % 64) return fail(nullptr, SONIC_ERR_INVALID, "vocab must be a multiple of 64");
    if (d.enc_T * 2 != d.n_frames || d.enc_T % d.merge || d.enc_T % 4) return fail(nullptr, SONIC_ERR_INVALID, "enc_T must be n_frames/2 and divisible by merge and 4");
    if ((2 * d.enc_d) % 128) return fail(nullptr, SONIC_ERR_INVALID, "2*enc_d must be a multiple of 128");
    if (max_batch < 1 || max_batch > 64) return fail(nullptr, SONIC_ERR_INVALID, "max_batch must be in 1..64");
    if (max_ctx < 64 || max_ctx % 64 || max_ctx > 8192) return fail(nullptr, SONIC_ERR_INVALID, "max_ctx must be a multiple of 64 in 64..8192");
    if (d.n_eos < 0 || d.n_eos > 8) return fail(nullptr, SONIC_ERR_INVALID, "n_eos must be 0..8");
    {   // every decode-step GEMM needs a weight-streaming tiling with 1..8 K slabs (a shape without one would silently produce zeros)
        const int QD = d.dec_heads * d.dec_head_dim, KD = d.dec_kv_heads * d.dec_head_dim;
        const int shapes[5][2] = {{QD + 2 * KD, d.dec_d}, {d.dec_d, QD}, {2 * d.dec_ff, d.dec_d}, {d.dec_d, d.dec_ff}, {d.vocab, d.dec_d}};
        const char* names[5] = {"qkv_proj", "o_proj", "gate/up_proj", "down_proj", "lm_head"};
        for (int i = 0; i < 5; ++i) {
            const int ks = (mode == SONIC_MODE_INT8 && i < 4) ? skinny_pick_ksplit_i8(shapes[i][0], shapes[i][1]) : skinny_pick_ksplit(shapes[i][0], shapes[i][1]);
            if (ks < 1 || ks > 8) return fail(nullptr, SONIC_ERR_INVALID, "decoder %s shape [%d x %d] has no decode-step tiling (K slabs = %d, need 1..8)", names[i], shapes[i][0], shapes[i][1], ks);
        }
    }
    return SONIC_OK;
}

extern "C" void sonic_destroy(sonic_engine* e);
// The main stream of a handle.  Experiments (tools/ab_stream_partition.sh, `make SONIC_AB=1` builds only): SONIC_EXP_PRIO / SONIC_EXP_CUS are comma lists indexed by the order in
// which this process created its handles - a stream priority (-1 high .. 1 low), or "lo-hi" = the CUs the handle's kernels may run on.
static hipError_t create_stream(hipStream_t* st) {
#ifdef SONIC_AB      // measured and lost (profiles/round4_stream_partition.txt): priorities change nothing, CU masks cost 19 .. 42 %
    static std::atomic<int> created{0};
    const int idx = created.fetch_add(1);
    auto field = [&](const char* env, char* out, size_t cap) -> bool {
        const char* v = getenv(env);
        if (!v) return false;
        for (int i = 0; i < idx && v; ++i) { v = strchr(v, ','); if (v) ++v; }
        if (!v || !*v || *v == ',') return false;
        size_t n = strcspn(v, ","); if (n >= cap) n = cap - 1;
        memcpy(out, v, n); out[n] = 0; return true;
    };
    char buf[64];
    if (field("SONIC_EXP_CUS", buf, sizeof buf)) {
        int lo = 0, hi = 255;
        if (sscanf(buf, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi >= lo && hi < 256) {
            uint32_t mask[8] = {0};
            for (int c = lo; c <= hi; ++c) mask[c / 32] |= 1u << (c % 32);
            return hipExtStreamCreateWithCUMask(st, 8, mask);
        }
    }
    if (field("SONIC_EXP_PRIO", buf, sizeof buf)) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, atoi(buf));
#endif
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
// Everything an engine (or a slot) owns besides weights and constants: its stream, PCM staging, activation buffers, KV cache, decode-step
static int f32_alloc(sonic_engine* e);
// buffers, control words, events.
static int alloc_state(sonic_engine* e) {
    if (hipSetDevice(e->device) != hipSuccess) { e->err = "hipSetDevice failed"; return SONIC_ERR_HIP; }
    // waits sleep instead of spinning (see stream_sync): the runtime's default (hipDeviceScheduleAuto) spins when it sees more CPUs than GPUs
    if (!getenv("SONIC_SPIN_SYNC") && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    if (create_stream(&e->st) != hipSuccess) { e->err = "hipStreamCreate failed"; return SONIC_ERR_HIP; }
    if (!getenv("SONIC_SPIN_SYNC") && hipEventCreateWithFlags(&e->sync_ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e->sync_ev = nullptr; }
    const sonic_dims& d = e->d;
    const int Bm = e->Bm, max_batch = e->Bm, max_ctx = e->max_ctx;
    e->T = d.enc_T; e->Tp = (d.enc_T + T_PAD_ALIGN - 1) / T_PAD_ALIGN * T_PAD_ALIGN; e->Ta = d.enc_T / d.merge; e->hd_e = d.enc_d / d.enc_heads;
    e->QD = d.dec_heads * d.dec_head_dim; e->KD = d.dec_kv_heads * d.dec_head_dim; e->qkvN = e->QD + 2 * e->KD;
    // prompt tokens of one batch: at most Ta audio rows per window plus text; a long max_ctx (multi-window requests) must not
    // multiply every prefill buffer by it
    { const int per = max_ctx < e->Ta + 256 ? max_ctx : e->Ta + 256; e->tok_cap = max_batch * per; }
    e->out_cap = max_ctx;
    const int T = e->T, C = d.enc_d;
    const size_t Mp = (size_t)Bm * T + 128;
    int s;
#define A(x) do { s = (x); if (s != SONIC_OK) return s; } while (0)
    A(dalloc(e, &e->pcm, (size_t)Bm * d.n_frames * 160)); A(dalloc(e, &e->n_samples_d, Bm)); A(dalloc(e, &e->ring_peak, Bm));
    A(dalloc(e, &e->logspec, (size_t)Bm * d.n_frames * d.n_mels)); A(dalloc(e, &e->segmax, Bm));
    A(dalloc(e, &e->feats_fm, (size_t)Bm * (d.n_frames + 2) * d.n_mels + 4096));
    A(dalloc(e, &e->h1, (size_t)Bm * (d.n_frames + 2) * C + 4096));
    A(dalloc(e, &e->x, Mp * C)); A(dalloc(e, &e->ln, Mp * C)); A(dalloc(e, &e->att, Mp * C));
    A(dalloc(e, &e->qk, Mp * 2 * C)); A(dalloc(e, &e->vt, (size_t)Bm * C * e->Tp)); A(dalloc(e, &e->ff, Mp * d.enc_ff));
    A(dalloc(e, &e->ph, ((size_t)Bm * e->Ta + 128) * 2 * d.dec_d)); A(dalloc(e, &e->pe, ((size_t)Bm * e->Ta + 128) * d.dec_d));
    const size_t tc = (size_t)e->tok_cap + 128;
    A(dalloc(e, &e->dx, tc * d.dec_d)); A(dalloc(e, &e->dhn, tc * d.dec_d)); A(dalloc(e, &e->dqkv, tc * e->qkvN));
    A(dalloc(e, &e->dq, tc * e->QD)); A(dalloc(e, &e->datt, tc * e->QD)); A(dalloc(e, &e->dact, tc * d.dec_ff));
    const size_t kvn = (size_t)d.dec_layers * Bm * d.dec_kv_heads * max_ctx * d.dec_head_dim;
    A(dalloc_big(e, &e->Kc, kvn)); A(dalloc_big(e, &e->Vc, kvn)); A(dalloc(e, &e->Vts, (size_t)Bm * d.dec_kv_heads * d.dec_head_dim * max_ctx));
    long mx = 2L * d.dec_ff; if (e->qkvN > mx) mx = e->qkvN; if (d.dec_d > mx) mx = d.dec_d;
    e->slabN = mx;
    A(dalloc_act(e, &e->ssq, (size_t)256 * 64));
    A(dalloc_act(e, &e->slab, (size_t)8 * 64 * mx)); A(dalloc_act(e, &e->lslab, (size_t)8 * 64 * d.vocab));
    A(dalloc_act(e, &e->slab2, (size_t)8 * 16 * d.dec_d)); A(dalloc_act(e, &e->sx2, (size_t)16 * d.dec_d));
    A(dalloc_act(e, &e->sx, (size_t)64 * d.dec_d)); A(dalloc_act(e, &e->shn, (size_t)64 * d.dec_d)); A(dalloc(e, &e->sq, (size_t)64 * e->QD));
    A(dalloc_act(e, &e->satt, (size_t)64 * e->QD)); A(dalloc_act(e, &e->sact, (size_t)64 * d.dec_ff));
    A(dalloc(e, &e->kv_len, 64)); A(dalloc(e, &e->tok_pos, 64)); A(dalloc(e, &e->n_new, 64)); A(dalloc(e, &e->finished, 64));
    A(dalloc(e, &e->max_new_d, 64)); A(dalloc(e, &e->n_active, 4)); A(dalloc(e, &e->out_ids, (size_t)64 * e->out_cap));
    A(dalloc(e, &e->step_ctr, 64)); A(dalloc(e, &e->seq_iota, 64));
    if (e->i8) {
        // widest Linear8bitLt input: encoder MLP (enc_ff), projector (enc_d * merge), decoder MLP (dec_ff)
        int kmax = d.enc_ff; if (C * d.merge > kmax) kmax = C * d.merge; if (d.dec_ff > kmax) kmax = d.dec_ff; if (2 * d.dec_d > kmax) kmax = 2 * d.dec_d;
        if (e->QD > kmax) kmax = e->QD;
        e->q_kmax = kmax;
        size_t rows = Mp > tc ? Mp : tc;
        size_t qa_bytes = Mp * (size_t)(d.enc_ff > C * d.merge ? d.enc_ff : C * d.merge);
        if (tc * (size_t)d.dec_ff > qa_bytes) qa_bytes = tc * (size_t)d.dec_ff;
        if (((size_t)Bm * e->Ta + 128) * 2 * d.dec_d > qa_bytes) qa_bytes = ((size_t)Bm * e->Ta + 128) * 2 * d.dec_d;
        A(dalloc(e, &e->qa, qa_bytes + 4096)); A(dalloc(e, &e->q_sca, rows)); A(dalloc(e, &e->q_flags, (size_t)64 * kmax + 64));
        A(dalloc(e, &e->q_oc_cnt, 64)); A(dalloc(e, &e->q_oc_list, (size_t)64 * kmax)); A(dalloc(e, &e->win_req, 64));
        A(dalloc(e, &e->qkv_rm, Mp * 3 * C));
        e->defer_cap = Mp * (size_t)C > tc * (size_t)d.dec_d ? Mp * (size_t)C : tc * (size_t)d.dec_d;
        A(dalloc(e, &e->defer_tmp, e->defer_cap, false));
        A(dalloc_act(e, &e->hn_q, (size_t)64 * d.dec_d)); A(dalloc_act(e, &e->att_q, (size_t)64 * e->QD)); A(dalloc_act(e, &e->act_q, (size_t)64 * d.dec_ff));
        A(dalloc(e, &e->sca_hn, 64)); A(dalloc(e, &e->sca_att, 64)); A(dalloc(e, &e->sca_act, 64));
        A(dalloc(e, &e->oc_hn, 64)); A(dalloc(e, &e->oc_att, 64)); A(dalloc(e, &e->oc_act, 64));
        A(dalloc(e, &e->ol_hn, (size_t)64 * d.dec_d)); A(dalloc(e, &e->ol_att, (size_t)64 * e->QD)); A(dalloc(e, &e->ol_act, (size_t)64 * d.dec_ff));
        A(dalloc(e, &e->ov_hn, (size_t)64 * d.dec_d)); A(dalloc(e, &e->ov_att, (size_t)64 * e->QD)); A(dalloc(e, &e->ov_act, (size_t)64 * d.dec_ff));
        A(dalloc(e, &e->amax_att, 64 * 4)); A(dalloc(e, &e->amax_act, 64 * 4)); A(dalloc(e, &e->big_att, 64 * 4));
    }
    A(dalloc(e, &e->src, tc)); A(dalloc(e, &e->tok_seq, tc)); A(dalloc(e, &e->tok_pos_pf, tc));
    A(dalloc(e, &e->q_off, 64)); A(dalloc(e, &e->q_len, 64)); A(dalloc(e, &e->last_row, 64));
    {
        int iota[64]; for (int i = 0; i < 64; ++i) iota[i] = i;
        if (h2d(e, e->seq_iota, iota, sizeof iota) != hipSuccess) { e->err = "memcpy failed"; return SONIC_ERR_HIP; }
    }
    if (hipHostMalloc((void**)&e->n_active_h, (CHK_RING + 1) * 4, hipHostMallocDefault) != hipSuccess) { e->err = "hipHostMalloc failed"; return SONIC_ERR_HIP; }
    if (hipHostMalloc((void**)&e->svc_h, (size_t)CHK_RING * SVC_WORDS * 4, hipHostMallocDefault) != hipSuccess) { e->err = "hipHostMalloc failed"; return SONIC_ERR_HIP; }
    // (the row-fetch stream st_io is created by the first sonic_service_begin: hardware queues are dealt to streams in creation order, and the main
    //  streams of an engine and its slots should take the first ones - see sonic_more_hw_queues)
    if (hipEventCreateWithFlags(&e->xfer_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->splice_ev, hipEventDisableTiming) != hipSuccess) { e->err = "hipEventCreate failed"; return SONIC_ERR_HIP; }
    e->plan_cap = 3 * tc + 8 * 64;
    for (int i = 0; i < 2; ++i) {
        if (hipHostMalloc((void**)&e->plan_buf[i], e->plan_cap * 4, hipHostMallocDefault) != hipSuccess) { e->err = "hipHostMalloc failed"; return SONIC_ERR_HIP; }
        if (hipEventCreateWithFlags(&e->plan_ev[i], (getenv("SONIC_SPIN_SYNC") ? 0 : hipEventBlockingSync) | hipEventDisableTiming) != hipSuccess) { e->err = "hipEventCreate failed"; return SONIC_ERR_HIP; }
    }
    e->plan_h = e->plan_buf[0];
    for (auto& v : e->ev) if (hipEventCreate(&v) != hipSuccess) { e->err = "hipEventCreate failed"; return SONIC_ERR_HIP; }
    e->gemm_ev.resize(8 * (size_t)(d.enc_layers > 0 ? d.enc_layers : 1));   // per layer: [start, end] of the QKV, o, fc1, fc2 GEMM launches
    for (auto& v : e->gemm_ev) if (hipEventCreate(&v) != hipSuccess) { e->err = "hipEventCreate failed"; return SONIC_ERR_HIP; }
#undef A
    for (auto& v : e->chk_ev) if (hipEventCreateWithFlags(&v, (getenv("SONIC_SPIN_SYNC") ? 0 : hipEventBlockingSync) | hipEventDisableTiming) != hipSuccess) { e->err = "hipEventCreate failed"; return SONIC_ERR_HIP; }
    e->n_samples_h.assign(Bm, 0);
    return SONIC_OK;
}

extern "C" int sonic_create(const sonic_dims* dims, int device_id, int mode, int max_batch, int max_ctx, sonic_engine** out) {
    if (!dims || !out) return fail(nullptr, SONIC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (mode != SONIC_MODE_NATIVE && mode != SONIC_MODE_INT8 && mode != SONIC_MODE_F16 && mode != SONIC_MODE_F32) return fail(nullptr, SONIC_ERR_INVALID, "mode must be either 'native' or 'int8'");
    g_opts = LaunchOpts{};
    TRY(check_dims(*dims, max_batch, max_ctx, mode));
    if (mode == SONIC_MODE_INT8 && (dims->dec_ff > 8192 || dims->enc_d % 128 || dims->enc_ff % 128 || (dims->dec_heads * dims->dec_head_dim) % 128))
        return fail(nullptr, SONIC_ERR_INVALID, "int8 mode: dec_ff must be <= 8192 and every quantised K a multiple of 128");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, SONIC_ERR_HIP, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, SONIC_ERR_INVALID, "device %d out of range (%d devices)", device_id, ndev);
    (void)probe_hw_queues(device_id);                     // once per device and process; warns when streams will alias (see sonic_runtime_info)
    sonic_engine* e = new sonic_engine();
    if (const char* v = getenv("SONIC_KEEP_ROWMAJOR")) e->opt_prefill_rowmajor = atoi(v) >= 2;   // =2: keep the row-major decoder weights AND read them (A/B of whole test runs)
    e->d = *dims; e->device = device_id; e->mode = mode; e->Bm = max_batch; e->max_ctx = max_ctx;
    e->i8 = mode == SONIC_MODE_INT8; e->dt = (e->i8 || mode == SONIC_MODE_F16) ? DT_F16 : DT_BF16;
    e->f32 = mode == SONIC_MODE_F32;
    if (e->f32) e->f = new F32State();
    int s = alloc_state(e);
    if (s == SONIC_OK && e->f32) s = f32_alloc(e);
    if (s == SONIC_OK) s = build_constants(e);
    if (s == SONIC_OK && stream_sync(e) != hipSuccess) { e->err = "stream sync failed"; s = SONIC_ERR_HIP; }
    if (s != SONIC_OK) { g_create_err = e->err; sonic_destroy(e); return s; }
    *out = e;
    return SONIC_OK;
}

// Another batch in flight on the SAME weights (what the reference's file mode does in spirit: backend/main.py:429-445 keeps three decodes in
// flight on one model object).  The slot is a full engine handle - stage / run / fetch / rings / options all work on it - with its own stream,
// activation buffers, KV cache, PCM staging and decode graphs; every weight and constant pointer is the owner's, so sonic_weight_bytes of the
// owner does not move and the slot's is 0.  The decode loop is latency-bound: a second batch's MFMA-bound encoder / prefill and its
// decode steps fill the bubbles of the first one's.  Slots die with their owner at the latest; sonic_destroy(slot) releases one early.
extern "C" int sonic_slot_create(sonic_engine* parent, sonic_engine** out) {
    if (!parent || !out) return fail(nullptr, SONIC_ERR_INVALID, "null argument");
    *out = nullptr;
    sonic_engine* root = parent->owner ? parent->owner : parent;
    std::lock_guard<std::mutex> lk(root->mu);
    (void)hipGetLastError();
    if (!root->finalized) return fail(nullptr, SONIC_ERR_INVALID, "sonic_slot_create needs an engine whose weights are finalized");
    if (root->f32) return fail(nullptr, SONIC_ERR_UNSUPPORTED, "SONIC_MODE_F32 is a test kind: no slots");
    g_opts = root->opts;
    sonic_engine* e = new sonic_engine();
    e->d = root->d; e->device = root->device; e->mode = root->mode; e->Bm = root->Bm; e->max_ctx = root->max_ctx; e->i8 = root->i8; e->dt = root->dt;
    int s = alloc_state(e);
    if (s == SONIC_OK && stream_sync(e) != hipSuccess) { e->err = "stream sync failed"; s = SONIC_ERR_HIP; }
    if (s != SONIC_OK) { g_create_err = e->err; sonic_destroy(e); return s; }
    // the owner's weights and constants, by pointer (read-only on the request path)
    e->conv1w = root->conv1w; e->conv2w = root->conv2w; e->conv1b = root->conv1b; e->conv2b = root->conv2b;
    e->enc = root->enc; e->enc_nw = root->enc_nw; e->enc_nb = root->enc_nb; e->gelu_lut = root->gelu_lut;
    e->pj1w = root->pj1w; e->pj2w = root->pj2w; e->pj1b = root->pj1b; e->pj2b = root->pj2b; e->qpj1 = root->qpj1; e->qpj2 = root->qpj2;
    e->embed = root->embed; e->embed_t = root->embed_t; e->dec = root->dec; e->dec_nw = root->dec_nw;
    e->lc = root->lc; e->enc_cs = root->enc_cs; e->dec_cs = root->dec_cs;
    e->opts = root->opts; e->opt_no_graph = root->opt_no_graph; e->opt_no_fused_rope = root->opt_no_fused_rope; e->opt_no_gelu_lut = root->opt_no_gelu_lut;
    e->opt_i8_defer_thr = root->opt_i8_defer_thr; e->opt_i8_no_xq = root->opt_i8_no_xq; e->opt_i8_no_lnq = root->opt_i8_no_lnq; e->opt_i8_no_qkv_fuse = root->opt_i8_no_qkv_fuse;
    e->opt_decode_chunk = root->opt_decode_chunk; e->opt_no_pre_norm = root->opt_no_pre_norm; e->opt_decode_gemv = root->opt_decode_gemv;
    e->weight_bytes = 0; e->finalized = true; e->owner = root;
    root->slots.push_back(e);
    *out = e;
    return SONIC_OK;
}
// what a caller that was handed engine pointers (sonic_pipeline_create) has to know about them: row capacity, context capacity, mode, device and
// the identity of the weight copy (the owner's address: equal for an engine and all of its slots)
extern "C" int sonic_engine_info(sonic_engine* e, int32_t* max_batch, int32_t* max_ctx, int32_t* mode, int32_t* device_id, const void** weights_id) {
    if (!e) return SONIC_ERR_INVALID;
    if (max_batch) *max_batch = e->Bm;
    if (max_ctx) *max_ctx = e->max_ctx;
    if (mode) *mode = e->mode;
    if (device_id) *device_id = e->device;
    if (weights_id) *weights_id = e->owner ? (const void*)e->owner : (const void*)e;
    return SONIC_OK;
}
extern "C" int sonic_slot_count(sonic_engine* e) {
    if (!e) return 0;
    sonic_engine* root = e->owner ? e->owner : e;
    std::lock_guard<std::mutex> lk(root->mu);
    return 1 + (int)root->slots.size();
}

static void ring_free(struct sonic_ring* r);
static void async_shutdown(sonic_engine* e);
extern "C" void sonic_destroy(sonic_engine* e) {
    if (!e) return;
    async_shutdown(e);                                   // the worker of sonic_run_staged_async finishes its batch and exits
    if (e->owner) {                                      // a slot leaves its owner's list (under the owner's lock: sonic_slot_create / sonic_slot_count walk it)
        std::lock_guard<std::mutex> lk(e->owner->mu);
        auto& v = e->owner->slots;
        v.erase(std::remove(v.begin(), v.end(), e), v.end());
    } else {
        std::vector<sonic_engine*> kids;
        { std::lock_guard<std::mutex> lk(e->mu); kids.swap(e->slots); }
        for (sonic_engine* k : kids) { k->owner = nullptr; k->finalized = false; sonic_destroy(k); }   // slots first: they read this engine's weights
    }
    (void)hipSetDevice(e->device);
    if (e->st) (void)stream_sync(e);
    if (e->splice_ev) {                                // a sibling that prefilled for this handle must not wait for an event that is about to be destroyed
        sonic_engine* root = e->owner ? e->owner : e;  // (the stream was just drained: every splice this handle queued has completed)
        std::lock_guard<std::mutex> lk(root->mu);
        if (root != e && root->wait_ev == e->splice_ev) root->wait_pending = false;
        for (sonic_engine* k : root->slots) if (k != e && k->wait_ev == e->splice_ev) k->wait_pending = false;
    }
    for (sonic_ring* r : e->rings) ring_free(r);       // rings the caller left behind go with their engine
    e->rings.clear();
    for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.second);
    for (void* p : e->allocs) (void)hipFree(p);
    if (e->dump) (void)hipFree(e->dump);
    if (e->force_d) (void)hipFree(e->force_d);
    if (e->taps) (void)hipFree(e->taps);
    if (e->feats_f32) (void)hipFree(e->feats_f32);
    if (e->n_active_h) (void)hipHostFree(e->n_active_h);
    for (int i = 0; i < 2; ++i) { if (e->plan_buf[i]) (void)hipHostFree(e->plan_buf[i]); if (e->plan_ev[i]) (void)hipEventDestroy(e->plan_ev[i]); }
    if (e->svc_h) (void)hipHostFree(e->svc_h);
    if (e->st_io) { (void)hipStreamSynchronize(e->st_io); (void)hipStreamDestroy(e->st_io); }
    if (e->xfer_ev) (void)hipEventDestroy(e->xfer_ev);
    if (e->splice_ev) (void)hipEventDestroy(e->splice_ev);
    for (auto& v : e->ev) if (v) (void)hipEventDestroy(v);
    for (auto& v : e->chk_ev) if (v) (void)hipEventDestroy(v);
    if (e->sync_ev) (void)hipEventDestroy(e->sync_ev);
    for (auto& v : e->gemm_ev) if (v) (void)hipEventDestroy(v);
    if (e->st) (void)hipStreamDestroy(e->st);
    delete e->f;                                        // (its device buffers were in e->allocs)
    delete e;
}

extern "C" int sonic_debug_ktrace(sonic_engine* e, int64_t* out, int64_t n) {
    if (!e || !out) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->kt) return fail(e, SONIC_ERR_INVALID, "ktrace is off");
    const int64_t have = (int64_t)8 * KT_SLOT_BLOCKS * 8;
    HIPC(e, stream_sync(e));
    HIPC(e, d2h(e, out, e->kt, (size_t)(n < have ? n : have) * 8));
    return SONIC_OK;
}
extern "C" const char* sonic_last_error(sonic_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }
extern "C" int64_t sonic_weight_bytes(sonic_engine* e) { return e ? e->weight_bytes : 0; }
extern "C" int sonic_synchronize(sonic_engine* e) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    HIPC(e, stream_sync(e));
    return SONIC_OK;
}

// what ASRModel.get_model_info() reads from torch.cuda (asr.py:501-506: version, device name, total memory) ...
extern "C" int sonic_device_info(int device_id, char* name, int name_cap, int64_t* total_bytes, int64_t* free_bytes, int32_t* hip_runtime_version) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) { (void)hipGetLastError(); return fail(nullptr, SONIC_ERR_INVALID, "device %d not available", device_id); }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device_id) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, SONIC_ERR_HIP, "hipGetDeviceProperties failed"); }
    if (name && name_cap > 0) { snprintf(name, (size_t)name_cap, "%s", pr.name[0] ? pr.name : pr.gcnArchName); }   // (some boxes of the pool report an empty marketing name)
    if (total_bytes) *total_bytes = (int64_t)pr.totalGlobalMem;
    if (free_bytes) {
        size_t fr = 0, tot = 0; int cur = 0;
        (void)hipGetDevice(&cur);
        if (hipSetDevice(device_id) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
        (void)hipSetDevice(cur);
        *free_bytes = (int64_t)fr;
    }
    if (hip_runtime_version) { int v = 0; if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); v = 0; } *hip_runtime_version = v; }
    return SONIC_OK;
}
// ... and what the debug dict of transcribe() reads from the caching allocator (asr.py:453-457): allocated = bytes in this handle's live
// device allocations (weights, activations, KV cache; rings excluded; a slot: its own buffers, the weights are its owner's); there is no
// caching layer under the engine, so reserved = allocated
extern "C" int sonic_memory_info(sonic_engine* e, int64_t* allocated_bytes, int64_t* reserved_bytes) {
    if (!e) return SONIC_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    if (allocated_bytes) *allocated_bytes = e->alloc_bytes;
    if (reserved_bytes) *reserved_bytes = e->alloc_bytes;
    return SONIC_OK;
}

// ------------------------------------------------------------------------------------------ weights
static int raw_alloc(sonic_engine* e, const std::string& name, const std::vector<int64_t>& shape, DevTensor** out) {
    DevTensor& t = e->raw[name];
    if (!t.p) {
        t.shape = shape; t.n = numel(shape);
        void* q = nullptr;
        HIPC(e, hipMalloc(&q, t.n * sizeof(bf16_t)));
        t.p = (bf16_t*)q; e->alloc_bytes += (int64_t)(t.n * sizeof(bf16_t));
    } else if (t.shape != shape) return fail(e, SONIC_ERR_INVALID, "tensor %s loaded twice with different shapes", name.c_str());
    *out = &t;
    return SONIC_OK;
}

extern "C" int sonic_load_tensor(sonic_engine* e, const char* name, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!e || !name || !data || !shape) return SONIC_ERR_INVALID;
    ENTER(e);
    if (e->finalized) return fail(e, SONIC_ERR_INVALID, "weights already finalized");
    std::vector<int64_t> shp(shape, shape + ndim);
    bool known = false;
    for (auto& it : inventory(e->d)) if (it.name == name) { known = true; if (it.shape != shp) return fail(e, SONIC_ERR_INVALID, "tensor %s: unexpected shape", name); }
    if (!known) return fail(e, SONIC_ERR_INVALID, "unknown tensor name %s", name);
    if (e->f32) {      // fp32 kind: the tensor stays fp32 (a bf16 source is widened exactly)
        const size_t n = numel(shp);
        float*& dst = e->f->raw[name];
        if (!dst) TRY(dalloc(e, &dst, n, false));
        if (dtype == SONIC_DTYPE_F32) HIPC(e, h2d(e, dst, data, n * 4));
        else if (dtype == SONIC_DTYPE_BF16) {
            bf16_t* tmp = nullptr;
            HIPC(e, hipMalloc((void**)&tmp, n * 2));
            hipError_t r = h2d(e, tmp, data, n * 2);
            if (r == hipSuccess) { launch_bf16_to_f32(tmp, dst, (long)n, e->st, DT_BF16); r = stream_sync(e); }
            (void)hipFree(tmp);
            HIPC(e, r);
        } else return fail(e, SONIC_ERR_INVALID, "dtype must be f32 or bf16");
        e->weight_bytes += (int64_t)n * 4;
        return SONIC_OK;
    }
    DevTensor* t;
    TRY(raw_alloc(e, name, shp, &t));
    if (dtype == SONIC_DTYPE_BF16) {
        HIPC(e, h2d(e, t->p, data, t->n * 2));
    } else if (dtype == SONIC_DTYPE_F32) {
        float* tmp = nullptr;
        HIPC(e, hipMalloc((void**)&tmp, t->n * 4));
        hipError_t r = h2d(e, tmp, data, t->n * 4);
        // int8 mode loads the checkpoint with torch_dtype=float16 (asr.py:156): an fp32 source goes straight to fp16
        if (r == hipSuccess) { launch_f32_to_bf16(tmp, t->p, (long)t->n, e->st, e->dt); r = stream_sync(e); if (e->dt == DT_F16) e->raw_f16[name] = true; }
        (void)hipFree(tmp);
        HIPC(e, r);
    } else return fail(e, SONIC_ERR_INVALID, "dtype must be f32 or bf16");
    return SONIC_OK;
}

extern "C" int sonic_load_synthetic(sonic_engine* e, uint64_t seed) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (e->finalized) return fail(e, SONIC_ERR_INVALID, "weights already finalized");
    for (auto& it : inventory(e->d)) {
        DevTensor* t = nullptr;
        float* t32 = nullptr;
        if (e->f32) { float*& dst = e->f->raw[it.name]; if (!dst) TRY(dalloc(e, &dst, numel(it.shape), false)); t32 = dst; e->weight_bytes += (int64_t)numel(it.shape) * 4; }
        else TRY(raw_alloc(e, it.name, it.shape, &t));
        float scale = 0.1f, offset = 0.f;
        if (it.kind == 0) { double fi = 1; for (size_t i = 1; i < it.shape.size(); ++i) fi *= (double)it.shape[i]; scale = (float)sqrt(3.0 / fi); }
        else if (it.kind == 1) scale = (float)sqrt(3.0 / (double)it.shape[1]);
        else if (it.kind == 3) offset = 1.0f;
        const uint64_t key = mix64h(seed * 0x9E3779B97F4A7C15ULL + fnv1a64h(it.name.c_str()));
        if (e->f32) launch_synth_fill(key, (long)numel(it.shape), scale, offset, nullptr, t32, e->st, e->opt_f32_synth_bf16);     // the generator's exact fp32 values (synth.py bf16=False); option f32_synth_bf16: the bf16-rounded ones (the weights of a bf16 engine with the same seed)
        else launch_synth_fill(key, (long)t->n, scale, offset, t->p, nullptr, e->st);
    }
    HIPC(e, stream_sync(e));
    return SONIC_OK;
}

static int need(sonic_engine* e, const std::string& name, DevTensor** t) {
    auto it = e->raw.find(name);
    if (it == e->raw.end() || !it->second.p) return fail(e, SONIC_ERR_INVALID, "missing weight tensor %s", name.c_str());
    *t = &it->second;
    return SONIC_OK;
}
static int to_f32(sonic_engine* e, const std::string& name, float** out) {
    DevTensor* t; TRY(need(e, name, &t));
    TRY(dalloc(e, out, t->n, false));
    launch_bf16_to_f32(t->p, *out, (long)t->n, e->st, e->dt);
    e->weight_bytes += (int64_t)t->n * 4;
    return SONIC_OK;
}
// int8 mode: row-wise int8 of a packed [N][K] fp16 matrix (Int8Params.cuda()); the 16-bit matrix is released afterwards
static int quantize(sonic_engine* e, bf16_t** w16, int N, int K, QW* q, bool tiled, bool kmajor = false) {
    TRY(dalloc(e, &q->cb, (size_t)N * K, false)); TRY(dalloc(e, &q->scb, (size_t)N, false));
    launch_quant_weights(*w16, q->cb, q->scb, N, K, e->st);
    e->weight_bytes += (int64_t)N * K + (int64_t)N * 4 - (int64_t)N * K * 2;
    if (tiled) {
        TRY(dalloc_big(e, &q->cbt, (size_t)N * K, false));
        launch_tile_weights_i8(q->cb, q->cbt, N, K, e->st);
        e->weight_bytes += (int64_t)N * K;
        // k-major copy for the decode consumers' outlier gathers (8 consecutive bytes per outlier column and 8 outputs; from the tiled copy the same 8 bytes lie in 8
        // different 16-byte pieces: 16 x the cache lines).  Rounds 3 - 5 kept it for all four decoder projections (1.29 GB at full size); round 6 keeps it only where it
        // pays - o_proj and down_proj, whose consumer (add + RMSNorm: one block per row walking the row's whole outlier list over 2048 outputs) got 25 % slower without it -
        // and lets the prefill epilogues, the side product, the attention prologue and SwiGLU gather from the tiled copy: 3 683 -> 2 865 MiB at the same step time.
        // SONIC_KEEP_CBK=1: all four (A/B); SONIC_NO_CBK=1: none (2 395 MiB, the 64-row step +4.9 %).
        if ((kmajor && !getenv("SONIC_NO_CBK")) || getenv("SONIC_KEEP_CBK")) {
            TRY(dalloc_big(e, &q->cbk, (size_t)N * K, false));
            launch_transpose_i8(q->cb, q->cbk, N, K, e->st);
            e->weight_bytes += (int64_t)N * K;
        }
    }
    HIPC(e, stream_sync(e));
    for (auto it = e->allocs.begin(); it != e->allocs.end(); ++it) if (*it == (void*)*w16) { e->allocs.erase(it); break; }
    (void)hipFree(*w16); *w16 = nullptr; e->alloc_bytes -= (int64_t)N * K * 2;
    if (q->cbt && !getenv("SONIC_KEEP_ROWMAJOR")) {
        // the row-major int8 matrix was the prefill GEMM's operand and its outlier-column source: the tiled and the k-major copy serve both now
        for (auto it = e->allocs.begin(); it != e->allocs.end(); ++it) if (*it == (void*)q->cb) { e->allocs.erase(it); break; }
        (void)hipFree(q->cb); q->cb = nullptr; q->cb_rowmajor_kept = false;
        e->alloc_bytes -= (int64_t)(((size_t)N * K + 3) / 4 * 4); e->weight_bytes -= (int64_t)N * K;
    }
    return SONIC_OK;
}
// concatenate row blocks of [rows_i][K] tensors
static int concat_rows(sonic_engine* e, const std::vector<std::string>& names, bf16_t** out) {
    size_t tot = 0; std::vector<DevTensor*> ts;
    for (auto& n : names) { DevTensor* t; TRY(need(e, n, &t)); ts.push_back(t); tot += t->n; }
    TRY(dalloc(e, out, tot, false));
    size_t o = 0;
    for (auto* t : ts) { HIPC(e, hipMemcpyAsync(*out + o, t->p, t->n * 2, hipMemcpyDeviceToDevice, e->st)); o += t->n; }
    e->weight_bytes += (int64_t)tot * 2;
    return SONIC_OK;
}
static int keep_raw(sonic_engine* e, const std::string& name, bf16_t** out) {
    DevTensor* t; TRY(need(e, name, &t));
    *out = t->p; e->allocs.push_back(t->p); t->p = nullptr;   // ownership moves to the engine's alloc list
    e->weight_bytes += (int64_t)t->n * 2;
    return SONIC_OK;
}

static int f32_finalize(sonic_engine* e);
extern "C" int sonic_finalize_weights(sonic_engine* e) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (e->finalized) return SONIC_OK;
    if (e->f32) return f32_finalize(e);
    const sonic_dims& d = e->d;
    const std::string at = "model.audio_tower.", pj = "model.multi_modal_projector.", lm = "model.language_model.";
    e->weight_bytes = 0;
    if (e->dt == DT_F16)   // a bf16 checkpoint (or the synthetic generator's bf16 values) loaded as fp16, in place (asr.py:156 torch_dtype=float16)
        for (auto& kv : e->raw) if (kv.second.p && !e->raw_f16.count(kv.first)) launch_bf16_to_f16(kv.second.p, kv.second.p, (long)kv.second.n, e->st);
    {   // conv stem in im2col order [C][3][Ci]
        DevTensor *w1, *w2; TRY(need(e, at + "conv1.weight", &w1)); TRY(need(e, at + "conv2.weight", &w2));
        TRY(dalloc(e, &e->conv1w, w1->n, false)); TRY(dalloc(e, &e->conv2w, w2->n, false));
        hipLaunchKernelGGL(conv_permute_kernel, dim3((w1->n + 255) / 256), dim3(256), 0, e->st, w1->p, e->conv1w, d.enc_d, d.n_mels);
        hipLaunchKernelGGL(conv_permute_kernel, dim3((w2->n + 255) / 256), dim3(256), 0, e->st, w2->p, e->conv2w, d.enc_d, d.enc_d);
        e->weight_bytes += (int64_t)(w1->n + w2->n) * 2;
        TRY(to_f32(e, at + "conv1.bias", &e->conv1b)); TRY(to_f32(e, at + "conv2.bias", &e->conv2b));
    }
    e->enc.resize(d.enc_layers);
    for (int i = 0; i < d.enc_layers; ++i) {
        const std::string p = at + "layers." + std::to_string(i) + ".";
        EncLayerW& L = e->enc[i];
        TRY(to_f32(e, p + "input_layernorm.weight", &L.ln1w)); TRY(to_f32(e, p + "input_layernorm.bias", &L.ln1b));
        TRY(concat_rows(e, {p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}, &L.wqkv));
        TRY(dalloc(e, &L.bqkv, (size_t)3 * d.enc_d, true));   // k_proj has no bias (modeling_glmasr.py:184)
        DevTensor *bq, *bv; TRY(need(e, p + "self_attn.q_proj.bias", &bq)); TRY(need(e, p + "self_attn.v_proj.bias", &bv));
        launch_bf16_to_f32(bq->p, L.bqkv, d.enc_d, e->st, e->dt);
        launch_bf16_to_f32(bv->p, L.bqkv + 2 * d.enc_d, d.enc_d, e->st, e->dt);
        TRY(keep_raw(e, p + "self_attn.o_proj.weight", &L.wo)); TRY(to_f32(e, p + "self_attn.o_proj.bias", &L.bo));
        TRY(to_f32(e, p + "post_attention_layernorm.weight", &L.ln2w)); TRY(to_f32(e, p + "post_attention_layernorm.bias", &L.ln2b));
        TRY(keep_raw(e, p + "mlp.fc1.weight", &L.w1)); TRY(to_f32(e, p + "mlp.fc1.bias", &L.b1));
        TRY(keep_raw(e, p + "mlp.fc2.weight", &L.w2)); TRY(to_f32(e, p + "mlp.fc2.bias", &L.b2));
        if (e->i8) {
            TRY(quantize(e, &L.wqkv, 3 * d.enc_d, d.enc_d, &L.qqkv, false)); TRY(quantize(e, &L.wo, d.enc_d, d.enc_d, &L.qo, false));
            TRY(quantize(e, &L.w1, d.enc_ff, d.enc_d, &L.q1, false)); TRY(quantize(e, &L.w2, d.enc_d, d.enc_ff, &L.q2, false));
        }
    }
    TRY(to_f32(e, at + "norm.weight", &e->enc_nw)); TRY(to_f32(e, at + "norm.bias", &e->enc_nb));
    TRY(keep_raw(e, pj + "linear_1.weight", &e->pj1w)); TRY(to_f32(e, pj + "linear_1.bias", &e->pj1b));
    TRY(keep_raw(e, pj + "linear_2.weight", &e->pj2w)); TRY(to_f32(e, pj + "linear_2.bias", &e->pj2b));
    if (e->i8) {   // both projector linears are swapped: the reference's skip pattern 'audio_proj' does not match 'multi_modal_projector'
        TRY(quantize(e, &e->pj1w, 2 * d.dec_d, d.enc_d * d.merge, &e->qpj1, false)); TRY(quantize(e, &e->pj2w, d.dec_d, 2 * d.dec_d, &e->qpj2, false));
    }
    TRY(keep_raw(e, lm + "embed_tokens.weight", &e->embed));
    e->dec.resize(d.dec_layers);
    for (int i = 0; i < d.dec_layers; ++i) {
        const std::string p = lm + "layers." + std::to_string(i) + ".";
        DecLayerW& L = e->dec[i];
        TRY(to_f32(e, p + "input_layernorm.weight", &L.ln1)); TRY(to_f32(e, p + "post_attention_layernorm.weight", &L.ln2));
        TRY(concat_rows(e, {p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight"}, &L.wqkv));
        TRY(keep_raw(e, p + "self_attn.o_proj.weight", &L.wo));
        // gate / up interleaved in 16-row groups (EPI_SWIGLU, swiglu_slab_kernel)
        DevTensor *g, *u; TRY(need(e, p + "mlp.gate_proj.weight", &g)); TRY(need(e, p + "mlp.up_proj.weight", &u));
        TRY(dalloc(e, &L.wgu, g->n * 2, false));
        const size_t blk = (size_t)16 * d.dec_d * 2;
        HIPC(e, hipMemcpy2DAsync(L.wgu, 2 * blk, g->p, blk, blk, d.dec_ff / 16, hipMemcpyDeviceToDevice, e->st));
        HIPC(e, hipMemcpy2DAsync((char*)L.wgu + blk, 2 * blk, u->p, blk, blk, d.dec_ff / 16, hipMemcpyDeviceToDevice, e->st));
        e->weight_bytes += (int64_t)g->n * 4;
        TRY(keep_raw(e, p + "mlp.down_proj.weight", &L.wdown));
        auto tiled = [&](const bf16_t* w, bf16_t** out, int N, int K) -> int {
            TRY(dalloc_big(e, out, (size_t)N * K, false));
            launch_tile_weights(w, *out, N, K, e->st);
            e->weight_bytes += (int64_t)N * K * 2;
            return SONIC_OK;
        };
        L.wgu_t8 = nullptr; L.wqkv_t = L.wo_t = L.wgu_t = L.wdown_t = nullptr;
        if (e->i8) {
            TRY(quantize(e, &L.wqkv, e->qkvN, d.dec_d, &L.qqkv, true)); TRY(quantize(e, &L.wo, d.dec_d, e->QD, &L.qo, true, true));
            TRY(quantize(e, &L.wgu, 2 * d.dec_ff, d.dec_d, &L.qgu, true)); TRY(quantize(e, &L.wdown, d.dec_d, d.dec_ff, &L.qdown, true, true));
            continue;
        }
        TRY(tiled(L.wqkv, &L.wqkv_t, e->qkvN, d.dec_d)); TRY(tiled(L.wo, &L.wo_t, d.dec_d, e->QD));
        TRY(tiled(L.wdown, &L.wdown_t, d.dec_d, d.dec_ff));
        if (skinny_gu_eligible(1, 2 * d.dec_ff, d.dec_d)) {          // fused gate/up kernel's layout (8-row gate/up interleave); both 16-bit element types (round 5)
            // ONE decode copy of gate/up: the unfused path (A/B, shapes the fused kernel does not take) multiplies the same tiles and its SwiGLU pass
            // reads the columns in the 8-row interleave (round 4 kept a second tiled copy in the 16-row interleave: 1.4 GB of the full-size model)
            TRY(dalloc_big(e, &L.wgu_t8, (size_t)2 * d.dec_ff * d.dec_d, false));
            launch_tile_weights_gu8(L.wgu, L.wgu_t8, 2 * d.dec_ff, d.dec_d, e->st);
            e->weight_bytes += (int64_t)2 * d.dec_ff * d.dec_d * 2;
        } else {
            TRY(tiled(L.wgu, &L.wgu_t, 2 * d.dec_ff, d.dec_d));
        }
    }
    if (!e->i8 && !getenv("SONIC_KEEP_ROWMAJOR")) {
        // the row-major decoder projections were only the prefill GEMMs' operand: those read the tiled copies now (GemmArgs.w_tiled)
        HIPC(e, stream_sync(e));
        auto drop = [&](bf16_t** w, size_t n) {
            if (!*w) return;
            for (auto it = e->allocs.begin(); it != e->allocs.end(); ++it) if (*it == (void*)*w) { e->allocs.erase(it); break; }
            (void)hipFree(*w); *w = nullptr; e->alloc_bytes -= (int64_t)((n * 2 + 3) / 4 * 4); e->weight_bytes -= (int64_t)n * 2;
        };
        for (auto& L : e->dec) {
            drop(&L.wqkv, (size_t)e->qkvN * d.dec_d); drop(&L.wo, (size_t)d.dec_d * e->QD);
            drop(&L.wgu, (size_t)2 * d.dec_ff * d.dec_d); drop(&L.wdown, (size_t)d.dec_d * d.dec_ff);
        }
    }
    TRY(dalloc_big(e, &e->embed_t, (size_t)d.vocab * d.dec_d, false));
    launch_tile_weights(e->embed, e->embed_t, d.vocab, d.dec_d, e->st);
    e->weight_bytes += (int64_t)d.vocab * d.dec_d * 2;
    TRY(to_f32(e, lm + "norm.weight", &e->dec_nw));
    HIPC(e, stream_sync(e));
    for (auto& kv : e->raw) if (kv.second.p) { (void)hipFree(kv.second.p); kv.second.p = nullptr; e->alloc_bytes -= (int64_t)kv.second.n * 2; }
    e->raw.clear();
    e->finalized = true;
    return SONIC_OK;
}

// ------------------------------------------------------------------------------------------ pipeline stages
static void gemm(sonic_engine* e, int epi, const bf16_t* A, long lda, const bf16_t* W, const float* bias, bf16_t* C, long ldc,
                 int M, int N, int K, const bf16_t* R = nullptr, long ldr = 0, int w_tiled = 0, int gu8 = 0) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.C = C; a.ldc = ldc; a.bias = bias; a.R = R; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dt = e->dt;
    a.w_tiled = w_tiled; a.gu8 = gu8;
    a.gelu_lut = e->opt_no_gelu_lut ? nullptr : e->gelu_lut;
    launch_gemm(a, epi, e->st);
}
// Rows -> reference-call groups for LLM.int8's outlier columns: group of row r = gmap ? gmap[r / gdiv] : r / gdiv, G groups
struct QGroup { const int* gmap; int gdiv; int G; };
// One nn.Linear of the model on X [M][K] (row stride ldx).  Native mode: the 16-bit GEMM.  int8 mode and a swapped module (q.cb):
// Linear8bitLt = activation quantisation (3 streaming passes) + int8 MFMA GEMM whose epilogue dequantises and adds the outlier columns.
// rope_cs (optional): fuse the encoder's partial RoPE on columns < rope_ncols into the epilogue; returns whether it was fused (only the
// 256x256 kernel does it), else the caller runs the separate RoPE pass.
static QuantActArgs make_qa(sonic_engine* e, const bf16_t* X, long ldx, int M, int K, const QGroup& grp) {
    QuantActArgs qa{};
    qa.X = X; qa.ld = ldx; qa.M = M; qa.K = K; qa.gmap = grp.gmap; qa.gdiv = grp.gdiv; qa.G = grp.G; qa.flags = e->q_flags;
    qa.q = e->qa; qa.sca = e->q_sca; qa.oc_cnt = e->q_oc_cnt; qa.oc_list = e->q_oc_list; qa.oc_ld = e->q_kmax;
    return qa;
}
// LayerNorm whose output feeds a Linear8bitLt directly (ld == K == d): in int8 mode the norm kernel also quantises the row
static bool layernorm_q(sonic_engine* e, const bf16_t* x, const float* w, const float* b, bf16_t* y, int M, int d, float eps, const QGroup& grp) {
    if (e->i8 && !e->opt_i8_no_lnq) {
        const QuantActArgs qa = make_qa(e, y, d, M, d, grp);
        launch_quant_act_begin(qa, e->st);
        launch_layernorm(x, w, b, y, M, d, eps, e->st, e->dt, &qa);
        return true;
    }
    launch_layernorm(x, w, b, y, M, d, eps, e->st, e->dt);
    return false;
}
static bool rmsnorm_q(sonic_engine* e, const bf16_t* x, const float* w, bf16_t* y, int M, int d, float eps, const QGroup& grp) {
    if (e->i8 && !e->opt_i8_no_lnq) {
        const QuantActArgs qa = make_qa(e, y, d, M, d, grp);
        launch_quant_act_begin(qa, e->st);
        launch_rmsnorm(x, w, y, M, d, eps, nullptr, e->st, e->dt, &qa);
        return true;
    }
    launch_rmsnorm(x, w, y, M, d, eps, nullptr, e->st, e->dt);
    return false;
}
// the encoder's fused q|k|v linear in int8 mode: where V^T goes when the 256x256 kernel can take the whole epilogue (RoPE on q / k on the way out
// of the staged tile, V transposed while staging)
struct QkvVt { bf16_t* Vt; int n_split, seg_T, vt_ld; long vt_seg_stride; };
static bool qlinear(sonic_engine* e, int epi, const bf16_t* X, long ldx, const bf16_t* w16, const QW& q, const float* bias, bf16_t* C, long ldc,
                    int M, int N, int K, const bf16_t* R, long ldr, const QGroup& grp, const float* rope_cs = nullptr, int rope_T = 0, int rope_ncols = 0,
                    bool prequant = false, const QkvVt* vt = nullptr) {
    if (!e->i8 || !(q.cb || q.cbt)) { gemm(e, epi, X, ldx, w16, bias, C, ldc, M, N, K, R, ldr); return false; }
    const QuantActArgs qa = make_qa(e, X, ldx, M, K, grp);
    if (prequant) launch_quant_act_finish(qa, e->st);      // the LayerNorm that wrote X also wrote its codes, absmax and flags
    else launch_quant_act(qa, e->st);
    GemmArgs a{};
    a.A = (const bf16_t*)e->qa; a.lda = K; a.W = (const bf16_t*)q.cb; a.C = C; a.ldc = ldc; a.bias = bias; a.R = R; a.ldr = ldr; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dt = DT_F16;
    // decoder projections: ONE int8 operand copy since round 5 - the fragment-tiled one of the decode step (+ the k-major copy for outlier columns)
    if (q.cbt && (!e->opt_prefill_rowmajor || !q.cb_rowmajor_kept)) { a.W = (const bf16_t*)q.cbt; a.w_tiled = 1; }     // (outlier columns are gathered from the tiled copy)
    a.q.sca = e->q_sca; a.q.scb = q.scb; a.q.x16 = X; a.q.ldx16 = ldx; a.q.oc_cnt = e->q_oc_cnt; a.q.oc_list = e->q_oc_list; a.q.oc_ld = e->q_kmax;
    a.q.row_group = grp.gmap; a.q.group_div = grp.gdiv;
    // int8 q|k|v: RoPE + V^T inside the GEMM's epilogue (round 3; the register form of the 16-bit kinds spilled beside the dequantisation, the int8
    // kind does both on the staged tile).  Returns true: the caller skips its RoPE and transpose passes.
    bool fuse = false;
    if (vt && rope_cs && !e->opt_i8_no_qkv_fuse && !g_opts.gemm_force128) {
        GemmArgs t = a; t.Vt = vt->Vt; t.n_split = vt->n_split; t.seg_T = vt->seg_T; t.vt_ld = vt->vt_ld; t.vt_seg_stride = vt->vt_seg_stride;
        if (gemm256_eligible(t, EPI_QKV_VT)) { a = t; a.rope_cs = rope_cs; a.rope_T = rope_T; a.rope_ncols = rope_ncols; epi = EPI_QKV_VT; fuse = true; }
    }
    // residual-epilogue linears (o_proj, fc2, down_proj: their inputs are activation outputs, where long outlier lists occur): requests with
    // more than `i8_defer_thr` outlier columns are finished by the dense side product instead of the epilogue's per-element list walk
    const bool defer = epi == EPI_BIAS_RESID && e->defer_tmp && e->opt_i8_defer_thr >= 0 && (size_t)M * ldc <= e->defer_cap;
    if (defer) { a.q.defer_out = e->defer_tmp; a.q.defer_thr = e->opt_i8_defer_thr; }
    launch_gemm(a, epi, e->st);
    if (defer) launch_i8_outlier_side(a, e->st);
    return fuse;
}

static int run_mel(sonic_engine* e, int W, bool want_f32) {
    const sonic_dims& d = e->d;
    if (want_f32 && !e->feats_f32) {
        HIPC(e, hipMalloc((void**)&e->feats_f32, (size_t)e->Bm * d.n_mels * d.n_frames * 4));
    }
    int mx = 0; for (int i = 0; i < W; ++i) mx = e->n_samples_h[i] > mx ? e->n_samples_h[i] : mx;
    launch_logmel(e->pcm, (long)d.n_frames * 160, e->n_samples_d, mx, e->lc, e->logspec, e->segmax, W, d.n_frames, d.n_mels,
                  e->feats_fm, want_f32 ? e->feats_f32 : nullptr, e->st, e->dt);
    return SONIC_OK;
}

// feats_fm (16-bit, frame-major, padded) -> pe [W*Ta][dec_d].  win_group: request of each window (int8 mode: LLM.int8 finds its outlier
// columns over all rows of one reference call, i.e. over all windows of a request - HF runs them as one encoder batch)
static int run_encoder(sonic_engine* e, int W, float* enc_layers_out, float* enc_out_host, int n_groups) {
    const sonic_dims& d = e->d;
    const int C = d.enc_d, T = e->T, M = W * T, H = d.enc_heads, dt = e->dt;
    const QGroup grp{e->win_req, T, n_groups}, grp_p{e->win_req, e->Ta, n_groups};
    {   // conv stem as two batched im2col-free GEMMs (modeling_glmasr.py:313-316); nn.Conv1d is not swapped in int8 mode
        GemmArgs a{};
        a.A = e->feats_fm; a.lda = d.n_mels; a.W = e->conv1w; a.bias = e->conv1b; a.C = e->h1 + C; a.ldc = C; a.dt = dt;
        a.M = d.n_frames; a.N = C; a.K = 3 * d.n_mels; a.batch = W;
        a.strideA = (long)(d.n_frames + 2) * d.n_mels; a.strideC = (long)(d.n_frames + 2) * C;
        launch_gemm(a, EPI_BIAS_GELU, e->st);
        GemmArgs b{};
        b.A = e->h1; b.lda = 2L * C; b.W = e->conv2w; b.bias = e->conv2b; b.C = e->x; b.ldc = C; b.dt = dt;
        b.M = T; b.N = C; b.K = 3 * C; b.batch = W; b.strideA = (long)(d.n_frames + 2) * C; b.strideC = (long)T * C;
        launch_gemm(b, EPI_BIAS_GELU, e->st);
    }
    float* tap = nullptr;
    if (enc_layers_out || enc_out_host) HIPC(e, hipMalloc((void**)&tap, (size_t)M * C * 4));
    e->gemm_ev_used = 0;
    for (int l = 0; l < d.enc_layers; ++l) {
        const EncLayerW& L = e->enc[l];
        const bool pq1 = layernorm_q(e, e->x, L.ln1w, L.ln1b, e->ln, M, C, d.enc_ln_eps, grp);
        const bool tev = e->opt_gemm_timing && (size_t)(8 * l + 7) < e->gemm_ev.size();
        auto mark = [&](int i) { if (tev) (void)hipEventRecord(e->gemm_ev[8 * l + i], e->st); };
        FlashArgs f{};
        f.dt = dt; f.T = T; f.Hq = H; f.Hkv = H; f.scale = 1.0f / sqrtf((float)e->hd_e); f.Vt = e->vt; f.vt_ld = e->Tp; f.O = e->att; f.o_ld = C;
        f.k_head_stride = e->hd_e; f.vt_seq_stride = (long)C * e->Tp; f.vt_head_stride = (long)e->hd_e * e->Tp;
        if (e->i8) {
            // Linear8bitLt q / k / v share their input, hence one quantisation and one fused int8 GEMM; Q | K | V land row-major
            // ([M][3C]) and V is transposed by its own pass (the fused V^T epilogue plus the dequantisation spills registers)
            mark(0);
            const bool can_fuse = e->hd_e == 64 && d.enc_rotary_dim == 32 && !e->opt_no_fused_rope;
            const QkvVt vt{e->vt, 2 * C, T, e->Tp, (long)C * e->Tp};
            const bool fused = qlinear(e, EPI_BIAS, e->ln, C, nullptr, L.qqkv, L.bqkv, e->qkv_rm, 3L * C, M, 3 * C, C, nullptr, 0, grp,
                                       can_fuse ? e->enc_cs : nullptr, T, 2 * C, pq1, &vt);
            mark(1);
            if (!fused) {
                launch_rope_enc(e->qkv_rm, 3L * C, M, T, 2 * H, e->hd_e, d.enc_rotary_dim, e->enc_cs, e->st, dt);
                launch_transpose_v(e->qkv_rm, 3L * C, 2 * C, e->vt, W, T, C, e->Tp, (long)C * e->Tp, e->st);
            }
            f.Q = e->qkv_rm; f.q_ld = 3L * C; f.K = e->qkv_rm + C; f.k_ld = 3L * C;
            f.q_seq_stride = (long)T * 3 * C; f.k_seq_stride = (long)T * 3 * C;
        } else {
            GemmArgs a{};
            a.A = e->ln; a.lda = C; a.W = L.wqkv; a.bias = L.bqkv; a.C = e->qk; a.ldc = 2L * C; a.M = M; a.N = 3 * C; a.K = C; a.batch = 1; a.dt = dt;
            a.Vt = e->vt; a.n_split = 2 * C; a.seg_T = T; a.vt_ld = e->Tp; a.vt_seg_stride = (long)C * e->Tp;
            // partial RoPE of q / k in the GEMM's epilogue (the rotation pairs are in one lane's accumulators): no separate HBM pass
            const bool roped = e->hd_e == 64 && d.enc_rotary_dim == 32 && !e->opt_no_fused_rope && !g_opts.gemm_force128 && gemm256_eligible(a, EPI_QKV_VT);
            if (roped) { a.rope_cs = e->enc_cs; a.rope_T = T; a.rope_ncols = 2 * C; }
            mark(0);
            launch_gemm(a, EPI_QKV_VT, e->st);
            mark(1);
            if (!roped) launch_rope_enc(e->qk, 2L * C, M, T, 2 * H, e->hd_e, d.enc_rotary_dim, e->enc_cs, e->st, dt);
            f.Q = e->qk; f.q_ld = 2L * C; f.K = e->qk + C; f.k_ld = 2L * C;
            f.q_seq_stride = (long)T * 2 * C; f.k_seq_stride = (long)T * 2 * C;
        }
        launch_flash(f, 64, false, W, T, e->st);
        mark(2);
        qlinear(e, EPI_BIAS_RESID, e->att, C, L.wo, L.qo, L.bo, e->x, C, M, C, C, e->x, C, grp);
        mark(3);
        const bool pq2 = layernorm_q(e, e->x, L.ln2w, L.ln2b, e->ln, M, C, d.enc_ln_eps, grp);
        mark(4);
        qlinear(e, EPI_BIAS_GELU, e->ln, C, L.w1, L.q1, L.b1, e->ff, d.enc_ff, M, d.enc_ff, C, nullptr, 0, grp, nullptr, 0, 0, pq2);
        mark(5);
        if (tev) e->gemm_ev_used = l + 1;
        mark(6);
        qlinear(e, EPI_BIAS_RESID, e->ff, d.enc_ff, L.w2, L.q2, L.b2, e->x, C, M, C, d.enc_ff, e->x, C, grp);
        mark(7);
        if (enc_layers_out) {
            launch_bf16_to_f32(e->x, tap, (long)M * C, e->st, dt);
            HIPC(e, stream_sync(e));
            for (int b = 0; b < W; ++b)
                HIPC(e, d2h(e, enc_layers_out + ((size_t)b * d.enc_layers + l) * T * C, tap + (size_t)b * T * C, (size_t)T * C * 4));
        }
    }
    launch_layernorm(e->x, e->enc_nw, e->enc_nb, e->ln, M, C, d.enc_ln_eps, e->st, dt);
    if (enc_out_host) {
        launch_bf16_to_f32(e->ln, tap, (long)M * C, e->st, dt);
        HIPC(e, stream_sync(e));
        HIPC(e, d2h(e, enc_out_host, tap, (size_t)M * C * 4));
    }
    if (tap) (void)hipFree(tap);
    // 4-frame merge is a view: [M][C] == [W*Ta][4C] (modeling_glmasr.py:392-397)
    const int Mp = W * e->Ta, PI = C * d.merge, PM = 2 * d.dec_d;
    qlinear(e, EPI_BIAS_GELU, e->ln, PI, e->pj1w, e->qpj1, e->pj1b, e->ph, PM, Mp, PM, PI, nullptr, 0, grp_p);
    qlinear(e, EPI_BIAS, e->ph, PM, e->pj2w, e->qpj2, e->pj2b, e->pe, d.dec_d, Mp, d.dec_d, PM, nullptr, 0, grp_p);
    return SONIC_OK;
}

static int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static int keep_rows(const sonic_dims& d, int n_valid_frames) {  // modeling_glmasr.py:399-403 (python floor division)
    int L = n_valid_frames;
    L = floordiv(L + 2 - 2 - 1, 1) + 1;
    L = floordiv(L + 2 - 2 - 1, 2) + 1;
    const int k = floordiv(L - d.merge, d.merge) + 1;
    const int Ta = d.enc_T / d.merge;
    return k < 0 ? 0 : (k > Ta ? Ta : k);
}
static int frames_of(int n_samples) { return n_samples > 0 ? (n_samples + 159) / 160 : 0; }

static long long* kt_slot(sonic_engine* e, int l, int slot) { return (e->kt && l == e->kt_layer) ? e->kt + (long)slot * KT_SLOT_BLOCKS * 8 : nullptr; }
static void skinny(sonic_engine* e, const bf16_t* X, long ldx, const bf16_t* W, float* P, int M, int N, int K, int* ks_out, long long* kt = nullptr) {
    SkinnyArgs a{}; a.kt = kt;
    a.X = X; a.ldx = ldx; a.W = W; a.P = P; a.M = M; a.N = N; a.K = K; a.ksplit = skinny_pick_ksplit(N, K); a.dt = e->dt;
    if (ks_out) *ks_out = a.ksplit;
    launch_skinny(a, e->st);
}
// int8 decode step: quantised rows Xq [M][K] x fragment-tiled int8 weights -> int32 slabs (dequantised by the consumer)
// the same projection on UNQUANTISED fp16 rows with their gathered absmax: the kernel quantises its X slice while staging it
static int skinny_i8_xq(sonic_engine* e, const bf16_t* X16, const float* amax, const int8_t* Wt, float* P, int M, int N, int K) {
    SkinnyArgs a{};
    a.X = X16; a.ldx = K; a.W = (const bf16_t*)Wt; a.P = P; a.M = M; a.N = N; a.K = K; a.ksplit = skinny_pick_ksplit_i8(N, K); a.i8 = 1; a.x_amax = amax;
    launch_skinny(a, e->st);
    return a.ksplit;
}
static int skinny_i8(sonic_engine* e, const int8_t* Xq, const int8_t* Wt, float* P, int M, int N, int K) {
    SkinnyArgs a{};
    a.X = (const bf16_t*)Xq; a.ldx = K; a.W = (const bf16_t*)Wt; a.P = P; a.M = M; a.N = N; a.K = K; a.ksplit = skinny_pick_ksplit_i8(N, K); a.i8 = 1;
    launch_skinny(a, e->st);
    return a.ksplit;
}

static GreedyArgs greedy_args(sonic_engine* e, int R, bool dump) {
    const sonic_dims& d = e->d;
    GreedyArgs g{};
    g.logits = e->lslab; g.ksplit = skinny_pick_ksplit(d.vocab, d.dec_d); g.mpad = ((R + 15) / 16) * 16; g.V = d.vocab; g.B = R; g.table = e->embed; g.x = e->sx; g.d = d.dec_d;
    g.out_ids = e->out_ids; g.out_ld = e->out_cap; g.n_new = e->n_new; g.finished = e->finished; g.kv_len = e->kv_len; g.tok_pos = e->tok_pos;
    g.max_new = e->max_new_d; g.n_active = e->n_active; g.dev_err = e->n_active + 1; g.n_eos = d.n_eos; g.pad_id = d.n_eos > 0 ? d.eos[0] : 0;
    for (int i = 0; i < d.n_eos; ++i) g.eos[i] = d.eos[i];
    g.logits_dump = dump ? e->dump : nullptr; g.dump_stride_step = (long)R * d.vocab; g.step_counter = dump ? e->step_ctr : nullptr;
    g.norm_w = e->dec.empty() ? nullptr : e->dec[0].ln1; g.norm_eps = d.dec_rms_eps; g.y = e->dec.empty() ? nullptr : e->shn;        // the next step's first RMSNorm rides along (d <= 8192)
    g.force_ids = e->force_d; g.force_ld = e->force_ld;
    g.dt = e->dt;
    if (e->i8) g.qo = QuantOut{e->hn_q, d.dec_d, e->sca_hn, e->oc_hn, e->ol_hn, d.dec_d, e->ov_hn};     // layer 0's q/k/v input, quantised
    return g;
}

// one decode step for R rows: sx ([R][d]) -> next token (generation/utils.py:2876-2943)
static void decode_step_i8(sonic_engine* e, int R, bool dump);
static void decode_step_f32(sonic_engine* e, int R, bool dump);
static void decode_step(sonic_engine* e, int R, bool dump) {
    if (e->f32) { decode_step_f32(e, R, dump); return; }
    if (e->i8) { decode_step_i8(e, R, dump); return; }
    const sonic_dims& d = e->d;
    const int D = d.dec_d, mpad = ((R + 15) / 16) * 16, dt = e->dt;
    if (e->opt_decode_gemv && R <= GEMV_MAX_ROWS && d.dec_layers > 0 && e->dec[0].wgu_t8 && gemv_eligible(R, e->qkvN, D, GEMV_SLAB_NORM) && gemv_eligible(R, D, e->QD, GEMV_RESID) &&
        gemv_eligible(R, 2 * d.dec_ff, D, GEMV_SWIGLU_NORM) && gemv_eligible(R, D, d.dec_ff, GEMV_RESID) && gemv_eligible(R, d.vocab, D, GEMV_SLAB_NORM)) {
        // 1 - 4 rows, opt-in (gemv.hip): q|k|v (norm inside) -> attention -> o_proj (+ residual) -> gate/up (norm inside, SwiGLU) -> down_proj (+ residual); lm_head (norm inside)
        e->step_launches_per_layer = 5;
        auto gv = [&](int mode, const bf16_t* X, long ldx, const bf16_t* W, int N, int K, const float* nw, float* P, bf16_t* resid, bf16_t* act) {
            GemvArgs g{}; g.X = X; g.ldx = ldx; g.W = W; g.M = R; g.N = N; g.K = K; g.dt = dt; g.norm_w = nw; g.eps = d.dec_rms_eps; g.P = P; g.resid = resid; g.ldr = D; g.act = act;
            launch_gemv(g, mode, e->st);
        };
        for (int l = 0; l < d.dec_layers; ++l) {
            const DecLayerW& L = e->dec[l];
            const size_t kvoff = (size_t)l * e->Bm * d.dec_kv_heads * e->max_ctx * d.dec_head_dim;
            gv(GEMV_SLAB_NORM, e->sx, D, L.wqkv_t, e->qkvN, D, L.ln1, e->slab, nullptr, nullptr);
            DecodeAttnArgs da{};
            da.P = e->slab; da.ksplit = 1; da.mpad = mpad; da.cs = e->dec_cs; da.dt = dt;
            da.Kc = e->Kc + kvoff; da.Vc = e->Vc + kvoff; da.O = e->satt; da.kv_len = e->kv_len; da.Hq = d.dec_heads; da.Hkv = d.dec_kv_heads;
            da.ctx_max = e->max_ctx; da.scale = 1.0f / sqrtf((float)d.dec_head_dim);
            launch_decode_attn(da, R, e->st);
            gv(GEMV_RESID, e->satt, e->QD, L.wo_t, D, e->QD, nullptr, nullptr, e->sx, nullptr);
            gv(GEMV_SWIGLU_NORM, e->sx, D, L.wgu_t8, 2 * d.dec_ff, D, L.ln2, nullptr, nullptr, e->sact);
            gv(GEMV_RESID, e->sact, d.dec_ff, L.wdown_t, D, d.dec_ff, nullptr, nullptr, e->sx, nullptr);
        }
        gv(GEMV_SLAB_NORM, e->sx, D, e->embed_t, d.vocab, D, e->dec_nw, e->lslab, nullptr, nullptr);
        GreedyArgs g = greedy_args(e, R, dump);
        g.ksplit = 1;
        launch_greedy(g, e->st);
        return;
    }
    int ks;
    if (D > 8192) launch_rmsnorm(e->sx, e->dec[0].ln1, e->shn, R, D, d.dec_rms_eps, nullptr, e->st, dt);   // else: done by the greedy kernel of the previous step
    // <= 2 rows (the B = 1 call shape of BASELINE configs 1 and 5; round 6): no standalone add+RMSNorm launch behind down_proj - the next q|k|v projection
    // (and, behind the last layer, the lm_head) sums the slabs, adds the residual and normalises its rows itself (skinny_xs_kernel<.., PRE>, the same
    // arithmetic statement by statement: same bits).  Five launches per layer instead of six.  down_proj then writes its slabs to slab2 (the consumer
    // writes slab while other blocks of it still read) and the residual stream alternates between sx and sx2.
    const bool fuse_all = !d.dec_layers ? false : (e->dec[0].wgu_t8 && skinny_gu_eligible(R, 2 * d.dec_ff, D) && skinny_o_eligible(R, D, e->QD));
    const bool pre = fuse_all && !e->opt_no_pre_norm && skinny_pre_eligible(R, e->qkvN, D) && skinny_pre_eligible(R, d.vocab, D) && D % 8 == 0;
    bf16_t* resid = e->sx;                               // where the residual rows live right now
    int ks_down = 0;
    e->step_launches_per_layer = !fuse_all ? 8 : pre ? 5 : ((e->opts.gu64_split_norm > 0 || (e->opts.gu64_split_norm == 0 && e->cap_svc)) && R > 32 && D % 128 == 0 && D <= 2048) ? 7 : 6;
    auto skinny_pre = [&](const bf16_t* W, float* P, int N, const float* nw, bf16_t* xout, int* ks_out, long long* kt) {
        SkinnyArgs a{}; a.kt = kt;
        a.X = resid; a.ldx = D; a.W = W; a.P = P; a.M = R; a.N = N; a.K = D; a.ksplit = skinny_pick_ksplit(N, D); a.dt = dt;
        a.pre_P = e->slab2; a.pre_ks = ks_down; a.pre_mpad = mpad; a.pre_x = resid; a.pre_xout = xout; a.pre_w = nw; a.pre_eps = d.dec_rms_eps;
        if (ks_out) *ks_out = a.ksplit;
        launch_skinny(a, e->st);
    };
    for (int l = 0; l < d.dec_layers; ++l) {
        const DecLayerW& L = e->dec[l];
        const size_t kvoff = (size_t)l * e->Bm * d.dec_kv_heads * e->max_ctx * d.dec_head_dim;
        if (pre && l > 0) {
            bf16_t* other = resid == e->sx ? e->sx2 : e->sx;
            skinny_pre(L.wqkv_t, e->slab, e->qkvN, L.ln1, other, &ks, kt_slot(e, l, 0));
            resid = other;
        } else
        skinny(e, e->shn, D, L.wqkv_t, e->slab, R, e->qkvN, D, &ks, kt_slot(e, l, 0));
        DecodeAttnArgs da{}; da.kt = kt_slot(e, l, 1);
        da.P = e->slab; da.ksplit = ks; da.mpad = mpad; da.cs = e->dec_cs; da.dt = dt;     // RoPE + KV append fused into the attention kernel
        da.Kc = e->Kc + kvoff; da.Vc = e->Vc + kvoff; da.O = e->satt; da.kv_len = e->kv_len; da.Hq = d.dec_heads; da.Hkv = d.dec_kv_heads;
        da.ctx_max = e->max_ctx; da.scale = 1.0f / sqrtf((float)d.dec_head_dim);
        if ((e->opts.decode_prefetch & 3) && R * d.dec_kv_heads <= 128 && L.wo_t) {
            // experiment: the attention launch covers R x Hkv of the 256 CUs - the others stream the weights of the kernels behind it
            da.pf_y = (256 - R * d.dec_kv_heads) / R;
            if (e->opts.decode_prefetch & 1) da.pf[0] = PrefetchRange{L.wo_t, (long)D * e->QD * 2};
            if ((e->opts.decode_prefetch & 2) && L.wgu_t8) da.pf[1] = PrefetchRange{L.wgu_t8, (long)d.dec_ff * D * 2};       // the first half of gate/up (25 MB)
        }
        launch_decode_attn(da, R, e->st);
        const bool fuse_gu = L.wgu_t8 && skinny_gu_eligible(R, 2 * d.dec_ff, D);
        const bool fuse_o = fuse_gu && skinny_o_eligible(R, D, e->QD);
        if (fuse_o) {
            // o_proj + residual add (+ row sum-of-squares partials) -> gate/up with RMSNorm applied while staging X + SwiGLU:
            // two kernels instead of o_proj, add+RMSNorm, gate/up, SwiGLU
            SkinnyArgs oa{}; oa.X = e->satt; oa.ldx = e->QD; oa.W = L.wo_t; oa.M = R; oa.N = D; oa.K = e->QD; oa.ksplit = 1; oa.dt = dt; oa.kt = kt_slot(e, l, 2);
            launch_skinny_o(oa, resid, D, e->ssq, e->st);
            SkinnyArgs ga{}; ga.X = resid; ga.ldx = D; ga.W = L.wgu_t8; ga.M = R; ga.N = 2 * d.dec_ff; ga.K = D; ga.ksplit = 1; ga.dt = dt; ga.kt = kt_slot(e, l, 3); ga.err = e->n_active + 1;
            if ((e->opts.gu64_split_norm > 0 || (e->opts.gu64_split_norm == 0 && e->cap_svc)) && R > 32 && D % 128 == 0 && D <= 2048) {
                // 33 .. 64 rows: the rows are normalised ONCE by their own small kernel (from the same partials, in the same order: same bits) and
                // gate/up stages them as they are - 256 blocks each normalising all 64 rows was the longest single piece of the 64-row step
                launch_rmsnorm_ss(e->sx, e->ssq, L.ln2, e->shn, R, D, d.dec_rms_eps, e->st, dt);
                ga.X = e->shn; launch_skinny_gu(ga, e->sact, e->st);
            } else
            launch_skinny_gu_norm(ga, e->sact, e->ssq, D / 16, L.ln2, d.dec_rms_eps, e->st);
        } else {
        skinny(e, e->satt, e->QD, L.wo_t, e->slab, R, D, e->QD, &ks);
        launch_add_rmsnorm(e->sx, e->slab, ks, mpad, L.ln2, e->shn, R, D, d.dec_rms_eps, e->st, dt);
        if (fuse_gu) {          // gate/up + SwiGLU in one kernel, no slabs
            SkinnyArgs ga{}; ga.X = e->shn; ga.ldx = D; ga.W = L.wgu_t8; ga.M = R; ga.N = 2 * d.dec_ff; ga.K = D; ga.ksplit = 1; ga.dt = dt;
            launch_skinny_gu(ga, e->sact, e->st);
        } else {
            skinny(e, e->shn, D, L.wgu_t ? L.wgu_t : L.wgu_t8, e->slab, R, 2 * d.dec_ff, D, &ks);
            launch_swiglu_slab(e->slab, ks, mpad, 2 * d.dec_ff, e->sact, R, e->st, dt, L.wgu_t ? 0 : 1);
        }
        }
        skinny(e, e->sact, d.dec_ff, L.wdown_t, pre ? e->slab2 : e->slab, R, D, d.dec_ff, &ks, kt_slot(e, l, 4));
        ks_down = ks;
        if (pre) continue;                               // the next layer's q|k|v (or the lm_head) consumes the slabs
        const float* nw = (l + 1 < d.dec_layers) ? e->dec[l + 1].ln1 : e->dec_nw;
        PrefetchRange pfq{nullptr, 0};
        if ((e->opts.decode_prefetch & 4) && l + 1 < d.dec_layers && e->dec[l + 1].wqkv_t) pfq = PrefetchRange{e->dec[l + 1].wqkv_t, (long)e->qkvN * D * 2};
        launch_add_rmsnorm(e->sx, e->slab, ks, mpad, nw, e->shn, R, D, d.dec_rms_eps, e->st, dt, nullptr, nullptr, &pfq, R < 256 ? 256 - R : 0);
    }
    if (pre) skinny_pre(e->embed_t, e->lslab, d.vocab, e->dec_nw, nullptr, nullptr, nullptr);   // tied lm_head behind the last layer's slabs (the updated residual is not needed again)
    else
    skinny(e, e->shn, D, e->embed_t, e->lslab, R, d.vocab, D, nullptr);   // tied lm_head (modeling_glmasr.py:517)
    launch_greedy(greedy_args(e, R, dump), e->st);
}

// int8 mode (asr.py:169-210): every decoder projection is a Linear8bitLt.  In a decode step each row is one reference call, so its
// outlier "columns" are its own elements >= 6.0.  Producers that own whole rows emit them quantised (greedy / add+RMSNorm / SwiGLU
// kernels; the attention output by its own one-block-per-row pass), the skinny int8 GEMM streams the fragment-tiled int8 weights
// (half the bytes of the bf16 step) into exact int32 slabs, and the consumer dequantises them (int8_util.h deq4).
static void decode_step_i8(sonic_engine* e, int R, bool dump) {
    const sonic_dims& d = e->d;
    e->step_launches_per_layer = 8;
    const int D = d.dec_d, FF = d.dec_ff, mpad = ((R + 15) / 16) * 16;
    const QuantOut q_hn{e->hn_q, D, e->sca_hn, e->oc_hn, e->ol_hn, D, e->ov_hn};
    const QuantOut q_att{e->att_q, e->QD, e->sca_att, e->oc_att, e->ol_att, e->QD, e->ov_att};
    const QuantOut q_act{e->act_q, FF, e->sca_act, e->oc_act, e->ol_act, FF, e->ov_act};
    auto deq = [&](const QuantOut& q, const QW& w, int K, const bf16_t* x16, int N) {
        DeqInfo dq{}; dq.sca = q.sca; dq.scb = w.scb; dq.cb = w.cb; dq.cbt = w.cbt; dq.cbk = w.cbk; dq.N = N; dq.K = K; dq.x16 = x16; dq.ldx16 = K; dq.oc_cnt = q.oc_cnt; dq.oc_list = q.oc_list; dq.oc_ld = q.oc_ld;
        dq.row_group = nullptr; dq.group_div = 1; dq.oc_val = q.oc_val; dq.dbg = e->opt_i8_dbg;
        return dq;
    };
    for (int l = 0; l < d.dec_layers; ++l) {
        const DecLayerW& L = e->dec[l];
        const size_t kvoff = (size_t)l * e->Bm * d.dec_kv_heads * e->max_ctx * d.dec_head_dim;
        int ks = skinny_i8(e, e->hn_q, L.qqkv.cbt, e->slab, R, e->qkvN, D);
        DecodeAttnArgs da{};
        da.P = e->slab; da.ksplit = ks; da.mpad = mpad; da.cs = e->dec_cs; da.dt = DT_F16; da.dq = deq(q_hn, L.qqkv, D, e->shn, e->qkvN);
        da.Kc = e->Kc + kvoff; da.Vc = e->Vc + kvoff; da.O = e->satt; da.kv_len = e->kv_len; da.Hq = d.dec_heads; da.Hkv = d.dec_kv_heads;
        da.ctx_max = e->max_ctx; da.scale = 1.0f / sqrtf((float)d.dec_head_dim);
        // o_proj's input rows are spread over the attention blocks of 4 kv heads: they gather the row absmax (atomicMax), o_proj quantises on the
        // fly and its consumer lists the outliers itself - no one-block-per-row quantisation launch in between (option i8_no_xq: the round-2 form)
        const bool xq = !e->opt_i8_no_xq && d.dec_kv_heads <= 4 && !e->opts.decode_attn_v1;   // the partials are [64][4]: one per kv-head block
        if (xq) { da.amax_out = e->amax_att; da.big_out = e->big_att; }
        launch_decode_attn(da, R, e->st);
        DeqInfo dq;
        if (xq) {
            ks = skinny_i8_xq(e, e->satt, e->amax_att, L.qo.cbt, e->slab, R, D, e->QD);
            dq = deq(q_att, L.qo, e->QD, e->satt, D); dq.sca = e->amax_att; dq.scan = 1; dq.scan_cnt = e->big_att;
        } else {
            launch_quant_rows(e->satt, e->QD, R, e->QD, q_att, e->st);
            ks = skinny_i8(e, e->att_q, L.qo.cbt, e->slab, R, D, e->QD);
            dq = deq(q_att, L.qo, e->QD, e->satt, D);
        }
        launch_add_rmsnorm(e->sx, e->slab, ks, mpad, L.ln2, e->shn, R, D, d.dec_rms_eps, e->st, DT_F16, &dq, &q_hn);
        ks = skinny_i8(e, e->hn_q, L.qgu.cbt, e->slab, R, 2 * FF, D);
        launch_swiglu_quant(e->slab, ks, mpad, FF, e->sact, R, deq(q_hn, L.qgu, D, e->shn, 2 * FF), q_act, e->st);
        ks = skinny_i8(e, e->act_q, L.qdown.cbt, e->slab, R, D, FF);
        dq = deq(q_act, L.qdown, FF, e->sact, D);
        const float* nw = (l + 1 < d.dec_layers) ? e->dec[l + 1].ln1 : e->dec_nw;
        launch_add_rmsnorm(e->sx, e->slab, ks, mpad, nw, e->shn, R, D, d.dec_rms_eps, e->st, DT_F16, &dq, &q_hn);
    }
    skinny(e, e->shn, D, e->embed_t, e->lslab, R, d.vocab, D, nullptr);   // lm_head is not swapped (asr.py:177): fp16 skinny GEMM
    launch_greedy(greedy_args(e, R, dump), e->st);
}

struct HostPlan {
    std::vector<int> src, tok_seq, tok_pos, q_off, q_len, last_row, max_new;
    int n_tok = 0, max_p = 0, max_steps = 0;
};

static int plan_requests(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                         const int32_t* max_new, HostPlan& hp) {
    const sonic_dims& d = e->d;
    const int W = e->W;
    if (R < 1 || R > e->Bm) return fail(e, SONIC_ERR_INVALID, "request count %d out of range 1..%d", R, e->Bm);
    if (!req_win && R != W) return fail(e, SONIC_ERR_INVALID, "R (%d) must equal staged windows (%d) when req_win is NULL", R, W);
    if (req_win && (req_win[0] != 0 || req_win[R] != W)) return fail(e, SONIC_ERR_INVALID, "req_win must cover exactly the staged windows");
    hp.q_off.resize(R); hp.q_len.resize(R); hp.last_row.resize(R); hp.max_new.resize(R);
    for (int r = 0; r < R; ++r) {
        const int w0 = req_win ? req_win[r] : r, w1 = req_win ? req_win[r + 1] : r + 1;
        if (w1 <= w0) return fail(e, SONIC_ERR_INVALID, "request %d has no audio window", r);
        std::vector<int> rows;  // audio rows of this request in order
        for (int w = w0; w < w1; ++w) {
            const int k = keep_rows(d, frames_of(e->n_samples_h[w]));
            for (int j = 0; j < k; ++j) rows.push_back(w * e->Ta + j);
        }
        const int64_t p0 = prompt_off[r], p1 = prompt_off[r + 1];
        const int P = (int)(p1 - p0);
        if (P < 1) return fail(e, SONIC_ERR_INVALID, "request %d has an empty prompt", r);
        if (max_new[r] < 1) return fail(e, SONIC_ERR_INVALID, "max_new_tokens must be >= 1");
        if (P + max_new[r] > e->max_ctx) return fail(e, SONIC_ERR_INVALID, "prompt (%d) + max_new_tokens (%d) exceeds max_ctx (%d)", P, max_new[r], e->max_ctx);
        if (max_new[r] > e->out_cap) return fail(e, SONIC_ERR_INVALID, "max_new_tokens too large");
        size_t used = 0; int n_ph = 0;
        for (int i = 0; i < P; ++i) n_ph += (prompt_ids[p0 + i] == d.audio_token_id);
        if ((size_t)n_ph != rows.size())
            return fail(e, SONIC_ERR_MISMATCH, "Audio features and audio tokens do not match, tokens: %d, features: %zu", n_ph, rows.size());
        hp.q_off[r] = hp.n_tok; hp.q_len[r] = P;
        for (int i = 0; i < P; ++i) {
            const int id = prompt_ids[p0 + i];
            if (id == d.audio_token_id) hp.src.push_back(-(1 + rows[used++]));
            else {
                if (id < 0 || id >= d.vocab) return fail(e, SONIC_ERR_INVALID, "token id %d out of vocabulary", id);
                hp.src.push_back(id);
            }
            hp.tok_seq.push_back(r); hp.tok_pos.push_back(i);
        }
        hp.n_tok += P;
        hp.last_row[r] = hp.n_tok - 1;
        hp.max_new[r] = max_new[r];
        if (P > hp.max_p) hp.max_p = P;
        if (max_new[r] > hp.max_steps) hp.max_steps = max_new[r];
    }
    if (hp.n_tok > e->tok_cap) return fail(e, SONIC_ERR_INVALID, "too many prompt tokens");
    return SONIC_OK;
}

static int run_prefill(sonic_engine* e, int R, const HostPlan& hp) {
    const sonic_dims& d = e->d;
    const int D = d.dec_d, M = hp.n_tok, dt = e->dt;
    const QGroup grp{e->tok_seq, 1, R};        // one reference call = the prompt rows of one request
    // The plan goes through pinned memory, so the copies are truly asynchronous and nothing here waits for the encoder that is still running
    // on this stream (a pageable source forced a stream synchronise between encoder and prefill: a host-dependent bubble in every batch).
    // plan_h = the staging buffer run_to_first_token picked for this run (free: the run that used it last has had its copies waited for).
    {
        int* h = e->plan_h; size_t o = 0;
        auto put = [&](int* dst, const int* srcv, size_t n) -> hipError_t {
            if (o + n > e->plan_cap) return hipErrorInvalidValue;
            memcpy(h + o, srcv, n * 4);
            hipError_t r = hipMemcpyAsync(dst, h + o, n * 4, hipMemcpyHostToDevice, e->st);
            o += n; return r;
        };
        HIPC(e, put(e->src, hp.src.data(), (size_t)M)); HIPC(e, put(e->tok_seq, hp.tok_seq.data(), (size_t)M)); HIPC(e, put(e->tok_pos_pf, hp.tok_pos.data(), (size_t)M));
        HIPC(e, put(e->q_off, hp.q_off.data(), (size_t)R)); HIPC(e, put(e->q_len, hp.q_len.data(), (size_t)R)); HIPC(e, put(e->kv_len, hp.q_len.data(), (size_t)R));
        HIPC(e, put(e->last_row, hp.last_row.data(), (size_t)R)); HIPC(e, put(e->max_new_d, hp.max_new.data(), (size_t)R));
        HIPC(e, put(e->n_active, &R, 1));
        HIPC(e, hipEventRecord(e->plan_ev[e->plan_idx], e->st));       // every copy out of this buffer (win_req included) is older than this event
        e->plan_busy[e->plan_idx] = true;
    }
    launch_fill_i32(e->n_new, 0, 64, e->st);
    if (e->amax_att) { launch_fill_i32((int*)e->amax_att, 0, 64 * 4, e->st); launch_fill_i32((int*)e->amax_act, 0, 64 * 4, e->st); launch_fill_i32(e->big_att, 0, 64 * 4, e->st); }   // (partials nobody writes stay 0)
    launch_fill_i32(e->finished, 0, 64, e->st);
    launch_fill_i32(e->step_ctr, 0, 64, e->st);
    launch_assemble_embeds(e->src, e->embed, e->pe, e->dx, M, D, e->st);
    e->last_ntok = M;
    if (e->taps_on) {
        if (!e->taps) HIPC(e, hipMalloc((void**)&e->taps, (size_t)(d.dec_layers + 1) * e->tok_cap * D * sizeof(bf16_t)));
        HIPC(e, hipMemcpyAsync(e->taps, e->dx, (size_t)M * D * 2, hipMemcpyDeviceToDevice, e->st));
    }
    for (int l = 0; l < d.dec_layers; ++l) {
        const DecLayerW& L = e->dec[l];
        const size_t kvoff = (size_t)l * e->Bm * d.dec_kv_heads * e->max_ctx * d.dec_head_dim;
        // 16-bit modes: the prefill GEMMs read the decode step's fragment-tiled weights (GemmArgs.w_tiled) - one copy of every decoder projection
        // since round 5 (the row-major ones are freed at load unless SONIC_KEEP_ROWMAJOR=1; option prefill_rowmajor then selects them for the A/B)
        auto lin = [&](int epi, const bf16_t* X, long ldx, const bf16_t* w16, const bf16_t* wt, int gu8, const QW& q, bf16_t* C, long ldc, int N, int K,
                       const bf16_t* Rr, long ldr, bool pq) {
            if (!e->i8 && wt && (!w16 || !e->opt_prefill_rowmajor)) gemm(e, epi, X, ldx, wt, nullptr, C, ldc, M, N, K, Rr, ldr, 1, gu8);
            else qlinear(e, epi, X, ldx, w16, q, nullptr, C, ldc, M, N, K, Rr, ldr, grp, nullptr, 0, 0, pq);
        };
        const bool pq1 = rmsnorm_q(e, e->dx, L.ln1, e->dhn, M, D, d.dec_rms_eps, grp);
        lin(EPI_BIAS, e->dhn, D, L.wqkv, L.wqkv_t, 0, L.qqkv, e->dqkv, e->qkvN, e->qkvN, D, nullptr, 0, pq1);
        RopeAppendArgs ra{}; ra.dt = dt;
        ra.qkv = e->dqkv; ra.ld = e->qkvN; ra.q_out = e->dq; ra.Kc = e->Kc + kvoff; ra.Vc = e->Vc + kvoff; ra.Vt = e->Vts; ra.vt_ld = e->max_ctx;
        ra.tok_seq = e->tok_seq; ra.tok_pos = e->tok_pos_pf; ra.cs = e->dec_cs; ra.Hq = d.dec_heads; ra.Hkv = d.dec_kv_heads; ra.ctx_max = e->max_ctx; ra.n_tok = M;
        if (!e->opt_no_rope_tiles) { ra.q_off = e->q_off; ra.q_len = e->q_len; ra.n_seq = R; ra.max_p = hp.max_p; }       // tiles of 16 positions per sequence (round 5)
        launch_rope_append(ra, false, e->st);
        FlashArgs f{}; f.dt = dt;
        f.Q = e->dq; f.q_ld = e->QD; f.K = e->Kc + kvoff; f.k_ld = d.dec_head_dim; f.Vt = e->Vts; f.vt_ld = e->max_ctx; f.O = e->datt; f.o_ld = e->QD;
        f.k_seq_stride = (long)d.dec_kv_heads * e->max_ctx * d.dec_head_dim; f.k_head_stride = (long)e->max_ctx * d.dec_head_dim;
        f.vt_seq_stride = (long)d.dec_kv_heads * d.dec_head_dim * e->max_ctx; f.vt_head_stride = (long)d.dec_head_dim * e->max_ctx;
        f.q_off = e->q_off; f.q_len = e->q_len; f.kv_len = e->q_len; f.Hq = d.dec_heads; f.Hkv = d.dec_kv_heads;
        f.scale = 1.0f / sqrtf((float)d.dec_head_dim);
        launch_flash(f, 128, true, R, hp.max_p, e->st);
        lin(EPI_BIAS_RESID, e->datt, e->QD, L.wo, L.wo_t, 0, L.qo, e->dx, D, D, e->QD, e->dx, D, false);
        const bool pq2 = rmsnorm_q(e, e->dx, L.ln2, e->dhn, M, D, d.dec_rms_eps, grp);
        lin(EPI_SWIGLU, e->dhn, D, L.wgu, L.wgu_t8 ? L.wgu_t8 : L.wgu_t, L.wgu_t8 ? 1 : 0, L.qgu, e->dact, d.dec_ff, 2 * d.dec_ff, D, nullptr, 0, pq2);
        lin(EPI_BIAS_RESID, e->dact, d.dec_ff, L.wdown, L.wdown_t, 0, L.qdown, e->dx, D, D, d.dec_ff, e->dx, D, false);
        if (e->taps_on) HIPC(e, hipMemcpyAsync(e->taps + (size_t)(l + 1) * e->tok_cap * D, e->dx, (size_t)M * D * 2, hipMemcpyDeviceToDevice, e->st));
    }
    // logits only for the last prompt position of each request (logits_to_keep=1, generation/utils.py:2612-2616)
    launch_rmsnorm(e->dx, e->dec_nw, e->shn, R, D, d.dec_rms_eps, e->last_row, e->st, dt);
    skinny(e, e->shn, D, e->embed_t, e->lslab, R, d.vocab, D, nullptr);
    return SONIC_OK;
}


// ------------------------------------------------------------------------------------------ SONIC_MODE_F32 (test only): fp32 stages, f32kind.hip
// The request plan (plan_requests), PCM staging, the log-mel kernel, the control words (kv_len / tok_pos / n_new / finished / out_ids / step logits dump /
// teacher forcing) and the greedy controller (greedy_kernel<float>) are the engine's own; what is different is the arithmetic between them.
static int f32_alloc(sonic_engine* e) {
    F32State& f = *e->f;
    const sonic_dims& d = e->d;
    const int Bm = e->Bm, C = d.enc_d;
    const size_t M = (size_t)Bm * e->T, tc = (size_t)e->tok_cap + 64;
    int s;
#define A(x) do { s = (x); if (s != SONIC_OK) return s; } while (0)
    A(dalloc(e, &f.featT, (size_t)Bm * (d.n_frames + 2) * d.n_mels + 4 * (size_t)d.n_mels)); A(dalloc(e, &f.h1, (size_t)Bm * (d.n_frames + 2) * C + 4 * (size_t)C));
    A(dalloc(e, &f.x, M * C)); A(dalloc(e, &f.ln, M * C)); A(dalloc(e, &f.q, M * C)); A(dalloc(e, &f.k, M * C)); A(dalloc(e, &f.v, M * C)); A(dalloc(e, &f.att, M * C));
    A(dalloc(e, &f.ff, M * d.enc_ff)); A(dalloc(e, &f.ph, (size_t)Bm * e->Ta * 2 * d.dec_d)); A(dalloc(e, &f.pe, (size_t)Bm * e->Ta * d.dec_d));
    A(dalloc(e, &f.dx, tc * d.dec_d)); A(dalloc(e, &f.dhn, tc * d.dec_d)); A(dalloc(e, &f.dq, tc * e->QD)); A(dalloc(e, &f.dk, tc * e->KD)); A(dalloc(e, &f.dv, tc * e->KD));
    A(dalloc(e, &f.datt, tc * e->QD)); A(dalloc(e, &f.dg, tc * d.dec_ff)); A(dalloc(e, &f.du, tc * d.dec_ff)); A(dalloc(e, &f.dact, tc * d.dec_ff));
    const size_t kvn = (size_t)d.dec_layers * Bm * e->max_ctx * e->KD;
    A(dalloc(e, &f.Kc, kvn)); A(dalloc(e, &f.Vc, kvn));
    A(dalloc(e, &f.logits, (size_t)64 * d.vocab)); A(dalloc(e, &f.hlast, (size_t)64 * d.dec_d));
#undef A
    return SONIC_OK;
}
static int f32_finalize(sonic_engine* e) {
    F32State& f = *e->f;
    const sonic_dims& d = e->d;
    const std::string at = "model.audio_tower.", pj = "model.multi_modal_projector.", lm = "model.language_model.";
    auto get = [&](const std::string& name, float** out) -> int {
        auto it = f.raw.find(name);
        if (it == f.raw.end() || !it->second) return fail(e, SONIC_ERR_INVALID, "missing weight tensor %s", name.c_str());
        *out = it->second;
        return SONIC_OK;
    };
    float *c1 = nullptr, *c2 = nullptr;
    TRY(get(at + "conv1.weight", &c1)); TRY(get(at + "conv2.weight", &c2));
    TRY(dalloc(e, &f.conv1w, (size_t)d.enc_d * d.n_mels * 3, false)); TRY(dalloc(e, &f.conv2w, (size_t)d.enc_d * d.enc_d * 3, false));
    launch_f32_conv_w(c1, f.conv1w, d.enc_d, d.n_mels, e->st); launch_f32_conv_w(c2, f.conv2w, d.enc_d, d.enc_d, e->st);   // [C][Ci][3] -> tap-major [C][3][Ci]
    TRY(get(at + "conv1.bias", &f.conv1b)); TRY(get(at + "conv2.bias", &f.conv2b));
    f.enc.resize(d.enc_layers);
    for (int i = 0; i < d.enc_layers; ++i) {
        const std::string p = at + "layers." + std::to_string(i) + ".";
        F32EncL& L = f.enc[i];
        TRY(get(p + "input_layernorm.weight", &L.ln1w)); TRY(get(p + "input_layernorm.bias", &L.ln1b));
        TRY(get(p + "self_attn.q_proj.weight", &L.wq)); TRY(get(p + "self_attn.q_proj.bias", &L.bq)); TRY(get(p + "self_attn.k_proj.weight", &L.wk));
        TRY(get(p + "self_attn.v_proj.weight", &L.wv)); TRY(get(p + "self_attn.v_proj.bias", &L.bv));
        TRY(get(p + "self_attn.o_proj.weight", &L.wo)); TRY(get(p + "self_attn.o_proj.bias", &L.bo));
        TRY(get(p + "post_attention_layernorm.weight", &L.ln2w)); TRY(get(p + "post_attention_layernorm.bias", &L.ln2b));
        TRY(get(p + "mlp.fc1.weight", &L.w1)); TRY(get(p + "mlp.fc1.bias", &L.b1)); TRY(get(p + "mlp.fc2.weight", &L.w2)); TRY(get(p + "mlp.fc2.bias", &L.b2));
    }
    TRY(get(at + "norm.weight", &f.enc_nw)); TRY(get(at + "norm.bias", &f.enc_nb));
    TRY(get(pj + "linear_1.weight", &f.pj1w)); TRY(get(pj + "linear_1.bias", &f.pj1b)); TRY(get(pj + "linear_2.weight", &f.pj2w)); TRY(get(pj + "linear_2.bias", &f.pj2b));
    TRY(get(lm + "embed_tokens.weight", &f.embed));
    f.dec.resize(d.dec_layers);
    for (int i = 0; i < d.dec_layers; ++i) {
        const std::string p = lm + "layers." + std::to_string(i) + ".";
        F32DecL& L = f.dec[i];
        TRY(get(p + "input_layernorm.weight", &L.ln1)); TRY(get(p + "self_attn.q_proj.weight", &L.wq)); TRY(get(p + "self_attn.k_proj.weight", &L.wk));
        TRY(get(p + "self_attn.v_proj.weight", &L.wv)); TRY(get(p + "self_attn.o_proj.weight", &L.wo)); TRY(get(p + "post_attention_layernorm.weight", &L.ln2));
        TRY(get(p + "mlp.gate_proj.weight", &L.wg)); TRY(get(p + "mlp.up_proj.weight", &L.wu)); TRY(get(p + "mlp.down_proj.weight", &L.wd));
    }
    TRY(get(lm + "norm.weight", &f.dec_nw));
    HIPC(e, stream_sync(e));
    e->finalized = true;
    return SONIC_OK;
}
static void f32_linear(sonic_engine* e, const float* X, long ldx, const float* W, const float* bias, float* Y, long ldy, int M, int N, int K, int epi = F32_EPI_NONE,
                       const float* R = nullptr, long ldr = 0) {
    F32Gemm g{};
    g.A = X; g.lda = ldx; g.W = W; g.C = Y; g.ldc = ldy; g.bias = bias; g.R = R; g.ldr = ldr; g.M = M; g.N = N; g.K = K; g.epi = epi;
    launch_f32_gemm(g, e->st);
}
// feats_f32 [W][n_mels][n_frames] -> pe [W * Ta][dec_d]   (modeling_glmasr.py:313-346, :380-408)
static int f32_run_encoder(sonic_engine* e, int W, float* enc_layers_out, float* enc_out_host) {
    F32State& f = *e->f;
    const sonic_dims& d = e->d;
    const int C = d.enc_d, T = e->T, M = W * T, H = d.enc_heads, hd = e->hd_e, NF = d.n_frames;
    launch_f32_feats_tm(e->feats_f32, f.featT, W, d.n_mels, NF, e->st);
    {   // conv stem: rows of the time-major padded input overlap (output t reads padded rows t .. t + 2; stride 2: 2t .. 2t + 2), taps-major weights
        F32Gemm a{};
        a.A = f.featT; a.lda = d.n_mels; a.sA1 = (long)(NF + 2) * d.n_mels; a.W = f.conv1w; a.bias = f.conv1b; a.C = f.h1 + C; a.ldc = C; a.sC1 = (long)(NF + 2) * C;
        a.M = NF; a.N = C; a.K = 3 * d.n_mels; a.nb1 = W; a.epi = F32_EPI_GELU;
        launch_f32_gemm(a, e->st);
        launch_f32_zero_pad_rows(f.h1, W, NF, C, e->st);
        F32Gemm b{};
        b.A = f.h1; b.lda = 2L * C; b.sA1 = (long)(NF + 2) * C; b.W = f.conv2w; b.bias = f.conv2b; b.C = f.x; b.ldc = C; b.sC1 = (long)T * C;
        b.M = T; b.N = C; b.K = 3 * C; b.nb1 = W; b.epi = F32_EPI_GELU;
        launch_f32_gemm(b, e->st);
    }
    for (int l = 0; l < d.enc_layers; ++l) {
        const F32EncL& L = f.enc[l];
        launch_f32_layernorm(f.x, L.ln1w, L.ln1b, f.ln, M, C, d.enc_ln_eps, e->st);
        f32_linear(e, f.ln, C, L.wq, L.bq, f.q, C, M, C, C);
        f32_linear(e, f.ln, C, L.wk, nullptr, f.k, C, M, C, C);                       // k_proj has no bias (modeling_glmasr.py:184)
        f32_linear(e, f.ln, C, L.wv, L.bv, f.v, C, M, C, C);
        launch_f32_rope(f.q, C, M, H, hd, d.enc_rotary_dim, e->enc_cs, nullptr, T, e->st);
        launch_f32_rope(f.k, C, M, H, hd, d.enc_rotary_dim, e->enc_cs, nullptr, T, e->st);
        F32Attn a{};
        a.Q = f.q; a.ldq = C; a.K = f.k; a.V = f.v; a.ldkv = C; a.seq_stride = (long)T * C; a.O = f.att; a.ldo = C; a.seq = nullptr; a.seq_div = T;
        a.pos = nullptr; a.lim_const = T; a.lim_max = T; a.hd = hd; a.grp = 1; a.scale = 1.0f / sqrtf((float)hd);
        launch_f32_attn(a, M, H, e->st);
        f32_linear(e, f.att, C, L.wo, L.bo, f.x, C, M, C, C, F32_EPI_RESID, f.x, C);
        launch_f32_layernorm(f.x, L.ln2w, L.ln2b, f.ln, M, C, d.enc_ln_eps, e->st);
        f32_linear(e, f.ln, C, L.w1, L.b1, f.ff, d.enc_ff, M, d.enc_ff, C, F32_EPI_GELU);
        f32_linear(e, f.ff, d.enc_ff, L.w2, L.b2, f.x, C, M, C, d.enc_ff, F32_EPI_RESID, f.x, C);
        if (enc_layers_out) {
            HIPC(e, stream_sync(e));
            for (int b = 0; b < W; ++b) HIPC(e, d2h(e, enc_layers_out + ((size_t)b * d.enc_layers + l) * T * C, f.x + (size_t)b * T * C, (size_t)T * C * 4));
        }
    }
    launch_f32_layernorm(f.x, f.enc_nw, f.enc_nb, f.ln, M, C, d.enc_ln_eps, e->st);
    if (enc_out_host) { HIPC(e, stream_sync(e)); HIPC(e, d2h(e, enc_out_host, f.ln, (size_t)M * C * 4)); }
    const int Mp = W * e->Ta, PI = C * d.merge, PM = 2 * d.dec_d;                       // the 4-frame merge is a view: [M][C] == [W * Ta][4C]
    f32_linear(e, f.ln, PI, f.pj1w, f.pj1b, f.ph, PM, Mp, PM, PI, F32_EPI_GELU);
    f32_linear(e, f.ph, PM, f.pj2w, f.pj2b, f.pe, d.dec_d, Mp, d.dec_d, PM);
    return SONIC_OK;
}
// the decoder layers over n_tok token rows of f.dx: token t belongs to sequence seq[t] and sits at position pos[t] (prefill: the prompt rows of all
// requests; token step: one row per request).  Keys / values are appended before the attention, which sees positions 0 .. pos[t] (llama:217-324)
static void f32_decoder_layers(sonic_engine* e, int n_tok, const int* seq, const int* pos) {
    F32State& f = *e->f;
    const sonic_dims& d = e->d;
    const int D = d.dec_d, QD = e->QD, KD = e->KD, hd = d.dec_head_dim, FF = d.dec_ff;
    const long seq_stride = (long)e->max_ctx * KD;
    for (int l = 0; l < d.dec_layers; ++l) {
        const F32DecL& L = f.dec[l];
        float* Kl = f.Kc + (size_t)l * e->Bm * seq_stride; float* Vl = f.Vc + (size_t)l * e->Bm * seq_stride;
        launch_f32_rmsnorm(f.dx, L.ln1, f.dhn, n_tok, D, d.dec_rms_eps, nullptr, e->st);
        f32_linear(e, f.dhn, D, L.wq, nullptr, f.dq, QD, n_tok, QD, D);
        f32_linear(e, f.dhn, D, L.wk, nullptr, f.dk, KD, n_tok, KD, D);
        f32_linear(e, f.dhn, D, L.wv, nullptr, f.dv, KD, n_tok, KD, D);
        launch_f32_rope(f.dq, QD, n_tok, d.dec_heads, hd, hd, e->dec_cs, pos, 0, e->st);
        launch_f32_rope(f.dk, KD, n_tok, d.dec_kv_heads, hd, hd, e->dec_cs, pos, 0, e->st);
        launch_f32_kv_append(f.dk, f.dv, Kl, Vl, seq, pos, n_tok, KD, seq_stride, e->st);
        F32Attn a{};
        a.Q = f.dq; a.ldq = QD; a.K = Kl; a.V = Vl; a.ldkv = KD; a.seq_stride = seq_stride; a.O = f.datt; a.ldo = QD; a.seq = seq; a.seq_div = 1;
        a.pos = pos; a.lim_const = 0; a.lim_max = e->max_ctx; a.hd = hd; a.grp = d.dec_heads / d.dec_kv_heads; a.scale = 1.0f / sqrtf((float)hd);
        launch_f32_attn(a, n_tok, d.dec_heads, e->st);
        f32_linear(e, f.datt, QD, L.wo, nullptr, f.dx, D, n_tok, D, QD, F32_EPI_RESID, f.dx, D);
        launch_f32_rmsnorm(f.dx, L.ln2, f.dhn, n_tok, D, d.dec_rms_eps, nullptr, e->st);
        f32_linear(e, f.dhn, D, L.wg, nullptr, f.dg, FF, n_tok, FF, D);
        f32_linear(e, f.dhn, D, L.wu, nullptr, f.du, FF, n_tok, FF, D);
        launch_f32_swiglu(f.dg, f.du, f.dact, (long)n_tok * FF, e->st);
        f32_linear(e, f.dact, FF, L.wd, nullptr, f.dx, D, n_tok, D, FF, F32_EPI_RESID, f.dx, D);
        if (e->taps_on && e->taps) (void)hipMemcpyAsync((float*)e->taps + (size_t)(l + 1) * e->tok_cap * D, f.dx, (size_t)n_tok * D * 4, hipMemcpyDeviceToDevice, e->st);
    }
}
static GreedyArgs f32_greedy_args(sonic_engine* e, int R, bool dump) {
    GreedyArgs g = greedy_args(e, R, dump);
    g.logits = e->f->logits; g.ksplit = 1; g.mpad = 64; g.table = (const bf16_t*)e->f->embed; g.x = (bf16_t*)e->f->dx; g.y = nullptr; g.norm_w = nullptr; g.dt = DT_F32;
    g.qo = QuantOut{};
    return g;
}
// final norm of the rows `last_row` (null: rows 0 .. R-1) + tied lm_head -> f.logits [R][vocab]
static void f32_lm_head(sonic_engine* e, int R, const int* last_row) {
    F32State& f = *e->f;
    const sonic_dims& d = e->d;
    launch_f32_rmsnorm(f.dx, f.dec_nw, f.hlast, R, d.dec_d, d.dec_rms_eps, last_row, e->st);
    f32_linear(e, f.hlast, d.dec_d, f.embed, nullptr, f.logits, d.vocab, R, d.vocab, d.dec_d);
}
static int f32_run_prefill(sonic_engine* e, int R, const HostPlan& hp) {
    F32State& f = *e->f;
    const sonic_dims& d = e->d;
    const int D = d.dec_d, M = hp.n_tok;
    {
        int* h = e->plan_h; size_t o = 0;
        auto put = [&](int* dst, const int* srcv, size_t n) -> hipError_t {
            memcpy(h + o, srcv, n * 4);
            hipError_t r = hipMemcpyAsync(dst, h + o, n * 4, hipMemcpyHostToDevice, e->st);
            o += n; return r;
        };
        HIPC(e, put(e->src, hp.src.data(), (size_t)M)); HIPC(e, put(e->tok_seq, hp.tok_seq.data(), (size_t)M)); HIPC(e, put(e->tok_pos_pf, hp.tok_pos.data(), (size_t)M));
        HIPC(e, put(e->q_off, hp.q_off.data(), (size_t)R)); HIPC(e, put(e->q_len, hp.q_len.data(), (size_t)R)); HIPC(e, put(e->kv_len, hp.q_len.data(), (size_t)R));
        HIPC(e, put(e->last_row, hp.last_row.data(), (size_t)R)); HIPC(e, put(e->max_new_d, hp.max_new.data(), (size_t)R));
        HIPC(e, put(e->n_active, &R, 1));
        HIPC(e, hipEventRecord(e->plan_ev[e->plan_idx], e->st));
        e->plan_busy[e->plan_idx] = true;
    }
    launch_fill_i32(e->n_new, 0, 64, e->st); launch_fill_i32(e->finished, 0, 64, e->st); launch_fill_i32(e->step_ctr, 0, 64, e->st);
    launch_f32_assemble(e->src, f.embed, f.pe, f.dx, M, D, e->st);
    e->last_ntok = M;
    if (e->taps_on) {
        if (!e->taps) HIPC(e, hipMalloc((void**)&e->taps, (size_t)(d.dec_layers + 1) * e->tok_cap * D * 4));
        HIPC(e, hipMemcpyAsync(e->taps, f.dx, (size_t)M * D * 4, hipMemcpyDeviceToDevice, e->st));
    }
    f32_decoder_layers(e, M, e->tok_seq, e->tok_pos_pf);
    f32_lm_head(e, R, e->last_row);            // logits of the last prompt position only (logits_to_keep = 1, generation/utils.py:2612-2616)
    return SONIC_OK;
}
// one token step for R rows (generation/utils.py:2876-2943): the rows' input embeddings are in f.dx (greedy_kernel<float> left them there)
static void decode_step_f32(sonic_engine* e, int R, bool dump) {
    f32_decoder_layers(e, R, e->seq_iota, e->tok_pos);
    f32_lm_head(e, R, nullptr);
    launch_greedy(f32_greedy_args(e, R, dump), e->st);
}

// log-mel -> encoder -> projector -> prefill -> first greedy token (generation/utils.py:2612-2616, 2876-2943 for the first step)
static int run_to_first_token(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                              const int32_t* max_new, bool want_logits) {
    const sonic_dims& d = e->d;
    if (!e->finalized) return fail(e, SONIC_ERR_INVALID, "weights not finalized");
    if (e->W < 1) return fail(e, SONIC_ERR_INVALID, "no PCM staged");
    if (e->svc_on) return fail(e, SONIC_ERR_INVALID, "this handle is decoding continuously (sonic_service_begin): prefill on another slot and splice the rows in");
    if (e->wait_pending) {                                  // rows of the previous batch were spliced into another handle: its copy kernels read this
        HIPC(e, hipStreamWaitEvent(e->st, e->wait_ev, 0));  // engine's KV cache and row state, which this run is about to overwrite
        e->wait_pending = false;
    }
    HostPlan hp;
    TRY(plan_requests(e, req_win, R, prompt_ids, prompt_off, max_new, hp));
    {   // staging buffer of this run: the one used two runs ago; its copies are almost always long done (a blocking wait otherwise)
        const int i = e->plan_idx ^ 1;
        if (e->plan_busy[i]) { HIPC(e, hipEventSynchronize(e->plan_ev[i])); e->plan_busy[i] = false; }
        e->plan_idx = i; e->plan_h = e->plan_buf[i];
    }
    if (e->force_d && (e->force_R != R || e->force_ld < hp.max_steps))
        return fail(e, SONIC_ERR_INVALID, "forced ids are [%d][%d] but the run has %d requests / %d steps", e->force_R, e->force_ld, R, hp.max_steps);
    e->R = R; e->max_steps = hp.max_steps; e->last_qlen = hp.q_len; e->last_maxnew = hp.max_new;
    if (want_logits) {
        const size_t need_n = (size_t)hp.max_steps * R * d.vocab;
        if (need_n > e->dump_cap) {
            if (e->dump) (void)hipFree(e->dump);
            e->dump = nullptr; e->dump_cap = 0;
            HIPC(e, hipMalloc((void**)&e->dump, need_n * 4));
            e->dump_cap = need_n;
        }
        e->dump_steps = hp.max_steps;
    } else e->dump_steps = 0;
    e->run_logits = want_logits;

    if (e->f32) {
        (void)hipEventRecord(e->ev[0], e->st);
        TRY(run_mel(e, e->W, true));
        (void)hipEventRecord(e->ev[1], e->st);
        TRY(f32_run_encoder(e, e->W, nullptr, nullptr));
        (void)hipEventRecord(e->ev[2], e->st);
        TRY(f32_run_prefill(e, R, hp));
        launch_greedy(f32_greedy_args(e, R, want_logits), e->st);
        (void)hipEventRecord(e->ev[3], e->st);
        e->steps_run = 0; e->greedy_calls = 1;
        return SONIC_OK;
    }
    (void)hipEventRecord(e->ev[0], e->st);
    TRY(run_mel(e, e->W, false));
    (void)hipEventRecord(e->ev[1], e->st);
    {   // window -> request map (int8 mode: outlier columns are found per request)
        std::vector<int> wr(64, 0);
        for (int r = 0; r < R; ++r) {
            const int w0 = req_win ? req_win[r] : r, w1 = req_win ? req_win[r + 1] : r + 1;
            for (int w = w0; w < w1 && w < 64; ++w) wr[w] = r;
        }
        if (e->i8) {                                           // (pinned: the last 64 words of plan_h, beyond what run_prefill uses)
            int* h = e->plan_h + e->plan_cap - 64;
            memcpy(h, wr.data(), 64 * 4);
            HIPC(e, hipMemcpyAsync(e->win_req, h, 64 * 4, hipMemcpyHostToDevice, e->st));
        }
    }
    TRY(run_encoder(e, e->W, nullptr, nullptr, R));
    (void)hipEventRecord(e->ev[2], e->st);
    TRY(run_prefill(e, R, hp));
    launch_greedy(greedy_args(e, R, want_logits), e->st);
    (void)hipEventRecord(e->ev[3], e->st);
    e->steps_run = 0;
    e->greedy_calls = 1;
    return SONIC_OK;
}

// A captured chunk of the greedy loop: `n` token steps for `R` rows as ONE hipGraph (kv_len / tok_pos / the token ids live on the device, so
// the steps of a chunk need nothing from the host).
// svc: the chunk belongs to a continuous decode loop (sonic_service_*), i.e. it runs beside a prefill slot and other loops by design - decode_step then
// picks the forms that cost the fewest CU-microseconds rather than the shortest chain (gu64_split_norm).  Same bits either way; cached separately.
static int chunk_graph(sonic_engine* e, int R, int n, hipGraphExec_t* out, bool svc = false) {
    const std::pair<int, int> key{R + (svc ? 4096 : 0), n};
    auto it = e->graphs.find(key);
    if (it != e->graphs.end()) { *out = it->second; return SONIC_OK; }
    hipGraph_t g = nullptr; hipGraphExec_t gx = nullptr;
    HIPC(e, hipStreamBeginCapture(e->st, hipStreamCaptureModeThreadLocal));
    e->cap_svc = svc;
    for (int i = 0; i < n; ++i) decode_step(e, R, false);
    e->cap_svc = false;
    HIPC(e, hipStreamEndCapture(e->st, &g));
    hipError_t r = hipGraphInstantiate(&gx, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIPC(e, r);
    e->graphs[key] = gx;
    *out = gx;
    return SONIC_OK;
}

// up to n_steps further token steps of the staged batch (HF:generation/utils.py:2876-2943), in chunks of `decode_chunk` steps: one hipGraph
// launch per chunk (eager under teacher forcing / when the step logits are wanted; a launch costs the host ~4 us, tools/host_cost.py).
// Ragged termination without a host round trip on the critical path: behind every chunk the device's count of running rows is copied to
// pinned memory and an event is recorded; the host reads check k only when chunk k + lookahead is already queued, so the stream never runs
// dry while the host looks, and the loop stops `lookahead` chunks after every row hit EOS / its budget (finished rows are frozen: the
// queued steps rewrite their own cache slot and emit nothing).  lookahead ADAPTS: it is 1 on a host that keeps up (waste at a stop: one
// chunk) and doubles whenever the host, about to launch the next chunk, finds the one it queued last already complete - the stream is empty,
// the device idle: a host that is descheduled for tens of milliseconds at a time (CPU quota shared with other work) or simply behind (a busy
// interpreter) - up to CHK_MAX_AHEAD chunks; a batch without such an observation takes one chunk off again.
static int run_decode_steps(sonic_engine* e, int n_steps, int* done_out) {
    const int R = e->R, left = e->max_steps - 1 - e->steps_run;
    if (n_steps > left) n_steps = left;
    const bool want_logits = e->run_logits;
    const bool use_graph = !want_logits && !e->opt_no_graph && !e->force_d && !e->f32;
    const int C = e->opt_decode_chunk > 0 ? e->opt_decode_chunk : 1;
    int done = 0;
    int launched = 0, checked = 0, last_grow = 0;   // chunks queued with a check behind them / checks the host has read
    bool all_stopped = false, starved = false, dev_err = false;
    typedef std::chrono::steady_clock clk;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    auto read_check = [&](bool block) -> int {  // 1: read (all_stopped updated), 0: not complete yet, < 0: error
        const int i = checked % CHK_RING;
        if (!block) { const hipError_t q = hipEventQuery(e->chk_ev[i]); if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; } if (q != hipSuccess) return -1; }
        else { const auto t_w = clk::now(); if (hipEventSynchronize(e->chk_ev[i]) != hipSuccess) return -1; e->host_wait_ms += ms_since(t_w); }
        if (e->n_active_h[i] <= 0) all_stopped = true;
        if (e->n_active_h[i] < DEV_ERR_ACTIVE) dev_err = true;
        ++checked;
        return 1;
    };
    while (done < n_steps && !all_stopped) {
        const int n = n_steps - done < C ? n_steps - done : C;
        // The chunk queued last is already complete as the next one is about to go out: the stream is empty, the device has been idle since -
        // this thread is late by more than a chunk (descheduled, or behind other host work).  Keep more chunks queued.
        if (launched > 0 && hipEventQuery(e->chk_ev[(launched - 1) % CHK_RING]) == hipSuccess) {
            if (launched - last_grow > e->lookahead && e->lookahead < CHK_MAX_AHEAD) {      // (the deeper queue gets `lookahead` launches to show before it grows again)
                e->lookahead = e->lookahead * 2 < CHK_MAX_AHEAD ? e->lookahead * 2 : CHK_MAX_AHEAD;
                last_grow = launched;
            }
            starved = true;
        }
        (void)hipGetLastError();
        const auto t_l = clk::now();
        if (use_graph) {
            hipGraphExec_t gx = nullptr;
            TRY(chunk_graph(e, R, n, &gx));
            HIPC(e, hipGraphLaunch(gx, e->st));
        } else {
            for (int i = 0; i < n; ++i) decode_step(e, R, want_logits);
        }
        e->host_launch_ms += ms_since(t_l); e->host_launches += 1;
        done += n; e->steps_run += n; e->greedy_calls += n;
        if (e->steps_run + 1 >= e->max_steps) break;           // the budget is exhausted: nothing left to stop early
        const int slot = launched % CHK_RING;
        HIPC(e, hipMemcpyAsync(e->n_active_h + slot, e->n_active, 4, hipMemcpyDeviceToHost, e->st));
        HIPC(e, hipEventRecord(e->chk_ev[slot], e->st));
        ++launched;
        (void)hipGetLastError();
        while (!all_stopped && checked < launched) {           // read what is there; block only for checks older than the lookahead
            const int r = read_check(launched - checked > e->lookahead);
            if (r < 0) return fail(e, SONIC_ERR_HIP, "decode loop: check event failed: %s", hipGetErrorString(hipGetLastError()));
            if (r == 0) break;
        }
    }
    e->run_starved = e->run_starved || starved;
    if (dev_err) return fail(e, SONIC_ERR_HIP, "a decode kernel gave up on an in-kernel wait: the batch's tokens are invalid");
    if (!all_stopped && done >= n_steps && e->steps_run + 1 < e->max_steps) {
        // the caller asked for fewer steps than the budget (sonic_decode_step) and synchronises next: read the outstanding checks now
        while (!all_stopped && checked < launched) if (read_check(true) < 0) return fail(e, SONIC_ERR_HIP, "decode loop: check event failed");
    }
    if (all_stopped) e->steps_run = e->max_steps - 1;
    if (done_out) *done_out = done;
    return SONIC_OK;
}

static int run_all(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                   const int32_t* max_new, bool want_logits) {
    const sonic_dims& d = e->d;
    const auto t_host0 = std::chrono::steady_clock::now();
    TRY(run_to_first_token(e, req_win, R, prompt_ids, prompt_off, max_new, want_logits));
    const double host_enqueue_first = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    e->host_launch_ms = e->host_wait_ms = 0; e->host_launches = 0; e->run_starved = false;
    int steps_done = 0;
    TRY(run_decode_steps(e, e->max_steps - 1, &steps_done));
    (void)hipEventRecord(e->ev[4], e->st);
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    sonic_timings& t = e->tim;
    memset(&t, 0, sizeof t);
    (void)hipEventElapsedTime(&t.mel_ms, e->ev[0], e->ev[1]);
    (void)hipEventElapsedTime(&t.encoder_ms, e->ev[1], e->ev[2]);
    (void)hipEventElapsedTime(&t.prefill_ms, e->ev[2], e->ev[3]);
    (void)hipEventElapsedTime(&t.decode_ms, e->ev[3], e->ev[4]);
    (void)hipEventElapsedTime(&t.total_ms, e->ev[0], e->ev[4]);
    for (int l = 0; l < e->gemm_ev_used; ++l) {
        float ms = 0; (void)hipEventElapsedTime(&ms, e->gemm_ev[8 * l + 4], e->gemm_ev[8 * l + 5]);
        t.gemm_ms += ms; t.gemm_launches += 1;
        const double MT = (double)e->W * e->T, C = d.enc_d, F = d.enc_ff;
        t.gemm_flops += 2.0 * MT * F * C;
        for (int g = 0; g < 4; ++g) { float m2 = 0; (void)hipEventElapsedTime(&m2, e->gemm_ev[8 * l + 2 * g], e->gemm_ev[8 * l + 2 * g + 1]); t.enc_gemm_ms += m2; }
        t.enc_gemm_flops += 2.0 * MT * C * (3 * C) + 2.0 * MT * C * C + 4.0 * MT * F * C;      // QKV, o, fc1, fc2
    }
    t.decode_steps = steps_done;
    t.host_prefill_enqueue_ms = (float)host_enqueue_first; t.host_decode_launch_ms = (float)e->host_launch_ms; t.host_decode_wait_ms = (float)e->host_wait_ms;
    t.host_decode_launches = e->host_launches; t.decode_lookahead = e->lookahead; t.decode_launches_per_layer = e->step_launches_per_layer;
    if (!e->run_starved && e->lookahead > 1) e->lookahead -= 1;       // a batch whose queue never ran dry: one chunk less ahead next time
    return SONIC_OK;
}

// ------------------------------------------------------------------------------------------ C ABI: hot path
static int stage_pcm_locked(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int W) {
    const sonic_dims& d = e->d;
    if (!pcm || !offsets) return fail(e, SONIC_ERR_INVALID, "null argument");
    if (W < 1 || W > e->Bm) return fail(e, SONIC_ERR_INVALID, "window count %d out of range 1..%d", W, e->Bm);
    const long cap = (long)d.n_frames * 160;
    for (int i = 0; i < W; ++i) {
        const int64_t n = offsets[i + 1] - offsets[i];
        if (n < 0 || n > cap) return fail(e, SONIC_ERR_INVALID, "window %d has %lld samples (max %ld)", i, (long long)n, cap);
        e->n_samples_h[i] = (int)n;
        if (n > 0) HIPC(e, hipMemcpyAsync(e->pcm + (size_t)i * cap, pcm + offsets[i], (size_t)n * 2, hipMemcpyHostToDevice, e->st));
    }
    HIPC(e, hipMemcpyAsync(e->n_samples_d, e->n_samples_h.data(), (size_t)W * 4, hipMemcpyHostToDevice, e->st));
    HIPC(e, stream_sync(e));
    e->W = W;
    return SONIC_OK;
}

static int fetch_locked(sonic_engine* e, int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits);
// ------------------------------------------------------------------------------------------ device-resident ingest
// A ring holds the raw wire PCM of one session in HBM (the reference keeps the chunks in a host dict, audio_manager.py:21-33, and
// concatenates them on the host for every decode, :106-123).  Appends run on the ring's own stream under the ring's own lock, so the
// event-loop thread that feeds 2048-byte chunks never waits for a batch that is decoding under the engine lock.
struct sonic_ring {
    sonic_engine* e = nullptr;
    int16_t* buf = nullptr;
    int16_t* host = nullptr;               // pinned mirror: an append is a host memcpy + an async H2D copy, the caller never waits for the
                                           // device (a synchronous 2 KB copy queues behind whatever kernels occupy the GPU: 0.6-1.4 ms measured)
    int64_t cap = 0, head = 0;             // capacity in samples; samples appended so far (absolute index of the next one)
    std::mutex mu;
    hipStream_t st = nullptr;
    hipEvent_t read_ev = nullptr; bool read_pending = false;   // last staging kernel that read this ring (appends order behind it)
    hipEvent_t app_ev = nullptr; bool app_pending = false;     // last append (staging kernels order behind it)
    int64_t unsynced = 0;                  // samples whose H2D copy may still be reading the pinned mirror
};

extern "C" int sonic_ring_create(sonic_engine* e, int64_t capacity_samples, sonic_ring** out) {
    if (!e || !out) return SONIC_ERR_INVALID;
    ENTER(e);
    if (capacity_samples < 1024 || capacity_samples > ((int64_t)1 << 31)) return fail(e, SONIC_ERR_INVALID, "ring capacity %lld out of range", (long long)capacity_samples);
    sonic_ring* r = new sonic_ring();
    r->e = e; r->cap = capacity_samples;
    if (hipMalloc((void**)&r->buf, (size_t)capacity_samples * 2) != hipSuccess) { delete r; return fail(e, SONIC_ERR_OOM, "HIP out of memory (ring of %lld samples)", (long long)capacity_samples); }
    if (hipHostMalloc((void**)&r->host, (size_t)capacity_samples * 2, hipHostMallocDefault) != hipSuccess) { (void)hipFree(r->buf); delete r; return fail(e, SONIC_ERR_OOM, "pinned host memory exhausted (ring mirror)"); }
    if (hipStreamCreateWithFlags(&r->st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&r->read_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->app_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(r->buf); (void)hipHostFree(r->host); if (r->st) (void)hipStreamDestroy(r->st); if (r->read_ev) (void)hipEventDestroy(r->read_ev);
        delete r; return fail(e, SONIC_ERR_HIP, "ring stream / event creation failed");
    }
    zero_fill(e, r->buf, (size_t)capacity_samples * 2);
    HIPC(e, stream_sync(e));
    // rings live in the registry of the weight owner: every slot of an engine may stage from every ring of it
    sonic_engine* root = e->owner ? e->owner : e;
    r->e = root;
    { std::lock_guard<std::mutex> rl(root->rings_mu); root->rings.push_back(r); }
    *out = r;
    return SONIC_OK;
}
static void ring_free(sonic_ring* r) {
    {
        std::lock_guard<std::mutex> lk(r->mu);
        (void)hipSetDevice(r->e->device);
        (void)hipStreamSynchronize(r->st);
        (void)hipFree(r->buf); (void)hipHostFree(r->host); (void)hipStreamDestroy(r->st); (void)hipEventDestroy(r->read_ev); (void)hipEventDestroy(r->app_ev);
    }
    delete r;
}
extern "C" void sonic_ring_destroy(sonic_ring* r) {
    if (!r) return;
    {
        // Unregister first: a batch that names this ring from now on is refused (stage_mixed_locked looks the pointer up under the same lock
        // before it touches it); a batch that already holds the ring's lock finishes its staging kernels before ring_free gets the lock.
        std::lock_guard<std::mutex> lk(r->e->rings_mu);
        auto& v = r->e->rings;
        v.erase(std::remove(v.begin(), v.end(), r), v.end());
    }
    ring_free(r);
}
extern "C" int64_t sonic_ring_head(sonic_ring* r) {
    if (!r) return -1;
    std::lock_guard<std::mutex> lk(r->mu);
    return r->head;
}
// append n samples; *first_index = absolute index of pcm[0].  Returns at once (the samples are copied to the pinned mirror, the caller may
// reuse pcm); the H2D copy is queued on the ring's stream and every later staging kernel orders behind it.
extern "C" int sonic_ring_append(sonic_ring* r, const int16_t* pcm, int64_t n, int64_t* first_index) {
    if (!r || (!pcm && n > 0) || n < 0) return SONIC_ERR_INVALID;
    std::lock_guard<std::mutex> lk(r->mu);
    if (n > r->cap) return SONIC_ERR_INVALID;
    // failures are reported through sonic_last_error(NULL) of the calling thread (appends do not take the engine lock, so they cannot
    // write the engine's own error string)
    auto hip_fail = [&](const char* what, hipError_t er) { (void)hipGetLastError(); return fail(nullptr, SONIC_ERR_HIP, "sonic_ring_append: %s failed: %s", what, hipGetErrorString(er)); };
    hipError_t er = hipSetDevice(r->e->device);
    if (er != hipSuccess) return hip_fail("hipSetDevice", er);
    // No device-side ordering against the staging kernels is needed: a batch holds the locks of its rings from the range check until its
    // staging kernels have COMPLETED (stage_mixed_locked ends with a stream synchronise), and this function runs under the ring's lock.
    // (Rounds 2-3 also recorded an event behind the staging kernels and made the ring's stream wait for it here.  The runtime refuses
    // both hipStreamWaitEvent and hipEventSynchronize on an event whose stream is capturing at that moment - the engine thread captures a
    // decode graph for every new batch size - "operation not permitted on an event last recorded in a capturing stream": an append then
    // failed, or left a sticky error that failed an unrelated call later.  Found by tests/test_gpu_sessions.py in full-suite runs.)
    const int64_t pos = r->head % r->cap, first = n < r->cap - pos ? n : r->cap - pos;
    // A mirror slot is rewritten one full capacity later (30 s of audio), normally long after its copy has left; the stream is
    // drained before an append could overwrite samples whose copy has not been waited for (small rings, bursts).
    if (r->unsynced + n > r->cap) { er = hipStreamSynchronize(r->st); if (er != hipSuccess) return hip_fail("hipStreamSynchronize", er); r->unsynced = 0; }
    r->unsynced += n;
    if (first > 0) {
        memcpy(r->host + pos, pcm, (size_t)first * 2);
        er = hipMemcpyAsync(r->buf + pos, r->host + pos, (size_t)first * 2, hipMemcpyHostToDevice, r->st);
        if (er != hipSuccess) return hip_fail("hipMemcpyAsync", er);
    }
    if (n > first) {
        memcpy(r->host, pcm + first, (size_t)(n - first) * 2);
        er = hipMemcpyAsync(r->buf, r->host, (size_t)(n - first) * 2, hipMemcpyHostToDevice, r->st);
        if (er != hipSuccess) return hip_fail("hipMemcpyAsync (wrap)", er);
    }
    if (n > 0) { er = hipEventRecord(r->app_ev, r->st); if (er != hipSuccess) return hip_fail("hipEventRecord", er); r->app_pending = true; }
    if (first_index) *first_index = r->head;
    r->head += n;
    return SONIC_OK;
}

// windows of a batch from host memory (rings == NULL or rings[w] == NULL: int16 PCM already normalised by the caller, as
// sonic_stage_pcm) and / or from rings (raw wire PCM: a1 + a2 on the device, peak over the windows of one request)
static int stage_mixed_locked(sonic_engine* e, int W, const int16_t* host_pcm, const int64_t* host_off, sonic_ring* const* rings,
                              const int64_t* ring_start, const int32_t* ring_n, const int32_t* req_win, int R) {
    const sonic_dims& d = e->d;
    if (W < 1 || W > e->Bm || W > RING_MAX_WIN) return fail(e, SONIC_ERR_INVALID, "window count %d out of range 1..%d", W, e->Bm < RING_MAX_WIN ? e->Bm : RING_MAX_WIN);
    if (req_win) { if (R < 1 || R > W || req_win[0] != 0 || req_win[R] != W) return fail(e, SONIC_ERR_INVALID, "req_win does not cover the %d windows", W); }
    else if (R != W) return fail(e, SONIC_ERR_INVALID, "without req_win every window is its own request");
    const long cap = (long)d.n_frames * 160;
    RingStageArgs ra{};
    int max_n = 0; bool any_ring = false;
    // every ring of the batch stays locked from the range check until the staging kernels have run (this function ends with a stream
    // synchronise): an append in between could overwrite the oldest samples of a window that starts at the tail of its ring
    std::vector<sonic_ring*> used;
    std::vector<std::unique_lock<std::mutex>> held;
    if (rings) {
        sonic_engine* root = e->owner ? e->owner : e;
        std::lock_guard<std::mutex> rl(root->rings_mu);        // registry lookup + ring locks as one step against sonic_ring_destroy
        for (int w = 0; w < W; ++w)
            if (rings[w] && std::find(used.begin(), used.end(), rings[w]) == used.end()) {
                if (std::find(root->rings.begin(), root->rings.end(), rings[w]) == root->rings.end())
                    return fail(e, SONIC_ERR_INVALID, "window %d: ring belongs to another engine (or was destroyed)", w);
                used.push_back(rings[w]);
            }
        std::sort(used.begin(), used.end());                   // one lock order for every batch (two slots may stage from overlapping ring sets)
        held.reserve(used.size());
        for (sonic_ring* rg : used) held.emplace_back(rg->mu);
    }
    for (int r = 0, w = 0; r < R; ++r) {
        const int w1 = req_win ? req_win[r + 1] : r + 1;
        if (w1 <= w) return fail(e, SONIC_ERR_INVALID, "request %d has no window", r);
        for (; w < w1; ++w) {
            ra.req_of[w] = r;
            sonic_ring* rg = rings ? rings[w] : nullptr;
            if (rg) {
                const int64_t n = ring_n[w], st = ring_start[w];
                if (n < 0 || n > cap || st < 0 || st + n > rg->head || st < rg->head - rg->cap)
                    return fail(e, SONIC_ERR_INVALID, "window %d: samples [%lld, %lld) are not in the ring (holds [%lld, %lld))", w, (long long)st, (long long)(st + n),
                                (long long)(rg->head > rg->cap ? rg->head - rg->cap : 0), (long long)rg->head);
                if (rg->app_pending && hipStreamWaitEvent(e->st, rg->app_ev, 0) != hipSuccess) {   // the appended samples are (or will be) in HBM first
                    (void)hipGetLastError();
                    HIPC(e, hipEventSynchronize(rg->app_ev));
                }
                ra.ring[w] = rg->buf; ra.ring_cap[w] = rg->cap; ra.start[w] = st % rg->cap; ra.n[w] = (int)n;
                e->n_samples_h[w] = (int)n;
                if ((int)n > max_n) max_n = (int)n;
                any_ring = true;
            } else {
                if (!host_pcm || !host_off) return fail(e, SONIC_ERR_INVALID, "window %d: neither ring nor host samples", w);
                const int64_t n = host_off[w + 1] - host_off[w];
                if (n < 0 || n > cap) return fail(e, SONIC_ERR_INVALID, "window %d has %lld samples (max %ld)", w, (long long)n, cap);
                e->n_samples_h[w] = (int)n;
                if (n > 0) HIPC(e, hipMemcpyAsync(e->pcm + (size_t)w * cap, host_pcm + host_off[w], (size_t)n * 2, hipMemcpyHostToDevice, e->st));
            }
        }
    }
    if (any_ring) {
        ra.peak = e->ring_peak; ra.pcm = e->pcm; ra.win_cap = cap;
        launch_fill_i32(e->ring_peak, 0, e->Bm, e->st);
        launch_ring_stage(ra, W, max_n, e->st);
    }
    HIPC(e, hipMemcpyAsync(e->n_samples_d, e->n_samples_h.data(), (size_t)W * 4, hipMemcpyHostToDevice, e->st));
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    e->W = W;
    return SONIC_OK;
}
extern "C" int sonic_stage_mixed(sonic_engine* e, const int16_t* host_pcm, const int64_t* host_off, sonic_ring* const* rings, const int64_t* ring_start,
                                 const int32_t* ring_n, int W, const int32_t* req_win, int R) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    return stage_mixed_locked(e, W, host_pcm, host_off, rings, ring_start, ring_n, req_win, R);
}
extern "C" int sonic_transcribe_mixed(sonic_engine* e, const int16_t* host_pcm, const int64_t* host_off, sonic_ring* const* rings, const int64_t* ring_start,
                                      const int32_t* ring_n, int W, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                                      const int32_t* max_new, int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits) {
    if (!e || !prompt_ids || !prompt_off || !max_new) return SONIC_ERR_INVALID;
    ENTER(e);
    TRY(stage_mixed_locked(e, W, host_pcm, host_off, rings, ring_start, ring_n, req_win, R));
    TRY(run_all(e, req_win, R, prompt_ids, prompt_off, max_new, step_logits != nullptr));
    return fetch_locked(e, out_ids, out_ld, out_len, step_logits);
}

extern "C" int sonic_stage_pcm(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int W) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    return stage_pcm_locked(e, pcm, offsets, W);
}

extern "C" int sonic_run_staged(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                                const int32_t* max_new, int want_step_logits) {
    if (!e || !prompt_ids || !prompt_off || !max_new) return SONIC_ERR_INVALID;
    ENTER(e);
    return run_all(e, req_win, R, prompt_ids, prompt_off, max_new, want_step_logits != 0);
}

// ---- continuous decoding (VERDICT r3 item 8 "row refill", generalised): one handle decodes forever over a pool of Bm rows, requests join and
// leave row by row.  The reference serialises every decode of every session (backend/connection_manager.py:127-245 awaits one transcribe() at a
// time per connection, backend/transcription_manager.py:58 blocks the event loop); a batch engine makes a request wait for the running batch to
// end and pads every batch to its slowest row.  Here a request is prefilled on ANOTHER handle of the same weights (sonic_prefill on a slot: log-mel,
// encoder, prompt forward, first token), its row - KV cache, control words, next-step input - is copied into a free row of the decoding handle
// between two chunks of its greedy loop, and the row is handed back the moment it hits EOS / its budget.  Rows are independent in every decode
// kernel and sum in a fixed order (batch invariance up to 32 rows), so a request's tokens are the same bits as in a solo run.
struct SpliceArgs {
    const bf16_t *Ks, *Vs; bf16_t *Kd, *Vd; int layers, Bm, Hkv, ctx, hd;
    int src[64], dst[64];
    const int *kv_len_s, *tok_pos_s, *n_new_s, *fin_s, *max_new_s, *out_s; int out_ld;
    int *kv_len_d, *tok_pos_d, *n_new_d, *fin_d, *max_new_d, *out_d, *n_active_d;
    const bf16_t *sx_s, *shn_s; bf16_t *sx_d, *shn_d; int D;
    const int8_t* hq_s; int8_t* hq_d; const float *sca_s, *ov_s; float *sca_d, *ov_d; const int *oc_s, *ol_s; int *oc_d, *ol_d;   // int8 mode: layer 0's quantised input row
};
__global__ __launch_bounds__(256) void splice_kv_kernel(SpliceArgs a) {
    const int i = blockIdx.x, lh = blockIdx.y, s = a.src[i], d = a.dst[i];
    const int len = min(max(a.kv_len_s[s], 1), a.ctx);
    const long blk = (long)a.ctx * a.hd;
    const long so = ((long)(lh / a.Hkv) * a.Bm + s) * a.Hkv + lh % a.Hkv, dd = ((long)(lh / a.Hkv) * a.Bm + d) * a.Hkv + lh % a.Hkv;
    const uint4* ks = (const uint4*)(a.Ks + so * blk); uint4* kd = (uint4*)(a.Kd + dd * blk);
    const uint4* vs = (const uint4*)(a.Vs + so * blk); uint4* vd = (uint4*)(a.Vd + dd * blk);
    const int n16 = len * a.hd / 8;
    for (int j = threadIdx.x; j < n16; j += 256) { kd[j] = ks[j]; vd[j] = vs[j]; }
}
__global__ __launch_bounds__(256) void splice_state_kernel(SpliceArgs a) {
    const int i = blockIdx.x, s = a.src[i], d = a.dst[i], t = threadIdx.x;
    for (int j = t; j < a.D / 8; j += 256) {
        ((uint4*)(a.sx_d + (long)d * a.D))[j] = ((const uint4*)(a.sx_s + (long)s * a.D))[j];
        ((uint4*)(a.shn_d + (long)d * a.D))[j] = ((const uint4*)(a.shn_s + (long)s * a.D))[j];
    }
    if (a.hq_s) {
        for (int j = t; j < a.D / 16; j += 256) ((uint4*)(a.hq_d + (long)d * a.D))[j] = ((const uint4*)(a.hq_s + (long)s * a.D))[j];
        const int cnt = a.oc_s[s];
        for (int j = t; j < cnt; j += 256) { a.ol_d[(long)d * a.D + j] = a.ol_s[(long)s * a.D + j]; a.ov_d[(long)d * a.D + j] = a.ov_s[(long)s * a.D + j]; }
        if (t == 0) { a.sca_d[d] = a.sca_s[s]; a.oc_d[d] = cnt; }
    }
    if (t == 0) {
        const int fin = a.fin_s[s];
        a.out_d[(long)d * a.out_ld] = a.out_s[(long)s * a.out_ld];            // the first token came out of the prefill
        a.kv_len_d[d] = a.kv_len_s[s]; a.tok_pos_d[d] = a.tok_pos_s[s]; a.n_new_d[d] = a.n_new_s[s]; a.max_new_d[d] = a.max_new_s[s];
        a.fin_d[d] = fin;
        if (!fin) atomicAdd(a.n_active_d, 1);
    }
}
// a fetched row goes back to the pool: it stays `finished` (frozen: it emits nothing) and looks at one key only until it is reused
__global__ void release_row_kernel(int* kv_len, int* tok_pos, int* finished, int row) { kv_len[row] = 1; tok_pos[row] = 0; finished[row] = 1; }
// the pipelined check of the continuous loop: finished | n_new | n_active -> one record in pinned host memory (a kernel's stores instead of
// three copy commands between every two chunks)
__global__ void service_status_kernel(const int* finished, const int* n_new, const int* n_active, int* out) {
    const int t = threadIdx.x;
    if (t < 64) { out[t] = finished[t]; out[64 + t] = n_new[t]; }
    if (t == 0) out[128] = *n_active;
}
__global__ void service_reset_kernel(int* kv_len, int* tok_pos, int* n_new, int* finished, int* max_new, int* n_active) {
    const int b = threadIdx.x;
    if (b < 64) { kv_len[b] = 1; tok_pos[b] = 0; n_new[b] = 0; finished[b] = 1; max_new[b] = 1; }
    if (b == 0) *n_active = 0;
}

extern "C" int sonic_service_begin(sonic_engine* e) {
    if (e && e->f32) return SONIC_ERR_UNSUPPORTED;
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->finalized) return fail(e, SONIC_ERR_INVALID, "weights not finalized");
    if (e->svc_on) return SONIC_OK;
    if (!e->st_io) HIPC(e, hipStreamCreateWithFlags(&e->st_io, hipStreamNonBlocking));
    HIPC(e, stream_sync(e));
    hipLaunchKernelGGL(service_reset_kernel, dim3(1), dim3(64), 0, e->st, e->kv_len, e->tok_pos, e->n_new, e->finished, e->max_new_d, e->n_active);
    if (e->force_d) return fail(e, SONIC_ERR_INVALID, "teacher forcing is set: clear it before continuous decoding");
    hipGraphExec_t gx = nullptr;                            // the chunk graphs exist before the first splice: nothing captures on this stream later
    if (e->Bm >= 2 && !e->i8) TRY(chunk_graph(e, 2, e->opt_decode_chunk > 0 ? e->opt_decode_chunk : 1, &gx, true));   // a pool with at most two occupied rows (round 6: five launches per layer)
    if (e->Bm >= 4 && !e->i8 && e->opt_decode_gemv) TRY(chunk_graph(e, 4, e->opt_decode_chunk > 0 ? e->opt_decode_chunk : 1, &gx, true));   // ... four, on the GEMV chain (opt-in)
    for (int R = 16; ; R += 16) {                           // one per 16 rows (sonic_service_step runs as many rows as are occupied)
        const int r = R < e->Bm ? R : e->Bm;
        TRY(chunk_graph(e, r, e->opt_decode_chunk > 0 ? e->opt_decode_chunk : 1, &gx, true));
        if (r >= e->Bm) break;
    }
    HIPC(e, stream_sync(e));
    e->svc_launched = e->svc_checked = 0; e->svc_seq = 0; e->svc_active = 0;
    for (int b = 0; b < 64; ++b) { e->svc_fin[b] = 1; e->svc_nn[b] = 0; }
    e->R = 0; e->greedy_calls = 0;
    e->svc_on = true;
    return SONIC_OK;
}
extern "C" int sonic_service_end(sonic_engine* e) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->svc_on) return SONIC_OK;
    HIPC(e, stream_sync(e));
    if (getenv("SONIC_SVC_STATS")) fprintf(stderr, "[sonic] continuous loop: %lld chunks queued, %lld launches found the stream empty with rows running, lookahead %d at the end\n",
                                           (long long)e->svc_launched, (long long)e->svc_dry, e->lookahead);
    e->svc_on = false;
    return SONIC_OK;
}
// n rows of `p` (requests 0 .. R-1 of its last sonic_prefill) -> rows dst_rows[] of the continuously decoding handle `d` (free rows: never
// spliced, or fetched since).  Queued on d's stream behind p's prefill; p's next run waits for the copies on the device.  *seq_out = chunks d
// had queued before the splice: checks with a larger sequence number (sonic_service_step) describe the new occupants of these rows.
extern "C" int sonic_splice_rows(sonic_engine* d, sonic_engine* p, int n, const int32_t* src_rows, const int32_t* dst_rows, int64_t* seq_out) {
    if (!d || !p || !src_rows || !dst_rows || d == p) return SONIC_ERR_INVALID;
    std::unique_lock<std::mutex> l1(d->mu, std::defer_lock), l2(p->mu, std::defer_lock);
    std::lock(l1, l2);
    (void)hipGetLastError();
    HIPC(d, hipSetDevice(d->device)); g_opts = d->opts;
    if (!d->svc_on) return fail(d, SONIC_ERR_INVALID, "sonic_splice_rows: the destination is not decoding continuously (sonic_service_begin)");
    if ((d->owner ? d->owner : d) != (p->owner ? p->owner : p)) return fail(d, SONIC_ERR_INVALID, "sonic_splice_rows: the handles do not share weights");
    if (d->Bm != p->Bm || d->max_ctx != p->max_ctx) return fail(d, SONIC_ERR_INVALID, "sonic_splice_rows: the handles differ in max_batch / max_ctx");
    if (n < 1 || n > 64 || p->greedy_calls < 1 || n > p->R) return fail(d, SONIC_ERR_INVALID, "sonic_splice_rows: %d rows, the source has %d prefilled requests", n, p->greedy_calls < 1 ? 0 : p->R);
    SpliceArgs a{};
    for (int i = 0; i < n; ++i) {
        if (src_rows[i] < 0 || src_rows[i] >= p->R || dst_rows[i] < 0 || dst_rows[i] >= d->Bm) return fail(d, SONIC_ERR_INVALID, "sonic_splice_rows: row out of range");
        for (int j = 0; j < i; ++j) if (dst_rows[j] == dst_rows[i]) return fail(d, SONIC_ERR_INVALID, "sonic_splice_rows: destination row %d named twice", dst_rows[i]);
        a.src[i] = src_rows[i]; a.dst[i] = dst_rows[i];
    }
    const sonic_dims& dm = d->d;
    a.Ks = p->Kc; a.Vs = p->Vc; a.Kd = d->Kc; a.Vd = d->Vc; a.layers = dm.dec_layers; a.Bm = d->Bm; a.Hkv = dm.dec_kv_heads; a.ctx = d->max_ctx; a.hd = dm.dec_head_dim;
    a.kv_len_s = p->kv_len; a.tok_pos_s = p->tok_pos; a.n_new_s = p->n_new; a.fin_s = p->finished; a.max_new_s = p->max_new_d; a.out_s = p->out_ids; a.out_ld = d->out_cap;
    a.kv_len_d = d->kv_len; a.tok_pos_d = d->tok_pos; a.n_new_d = d->n_new; a.fin_d = d->finished; a.max_new_d = d->max_new_d; a.out_d = d->out_ids; a.n_active_d = d->n_active;
    a.sx_s = p->sx; a.shn_s = p->shn; a.sx_d = d->sx; a.shn_d = d->shn; a.D = dm.dec_d;
    if (d->i8) { a.hq_s = p->hn_q; a.hq_d = d->hn_q; a.sca_s = p->sca_hn; a.sca_d = d->sca_hn; a.oc_s = p->oc_hn; a.oc_d = d->oc_hn; a.ol_s = p->ol_hn; a.ol_d = d->ol_hn; a.ov_s = p->ov_hn; a.ov_d = d->ov_hn; }
    HIPC(d, hipEventRecord(p->xfer_ev, p->st));
    HIPC(d, hipStreamWaitEvent(d->st, p->xfer_ev, 0));
    hipLaunchKernelGGL(splice_kv_kernel, dim3(n, dm.dec_layers * dm.dec_kv_heads), dim3(256), 0, d->st, a);
    hipLaunchKernelGGL(splice_state_kernel, dim3(n), dim3(256), 0, d->st, a);
    HIPC(d, hipEventRecord(d->splice_ev, d->st));
    // (rows of one prefill may go to several decoders: an earlier splice's event is not forgotten, the source's stream takes it on now)
    if (p->wait_pending && p->wait_ev != d->splice_ev) HIPC(d, hipStreamWaitEvent(p->st, p->wait_ev, 0));
    p->wait_ev = d->splice_ev; p->wait_pending = true;
    HIPC(d, hipGetLastError());
    if (seq_out) *seq_out = d->svc_launched;
    return SONIC_OK;
}
// queue n_chunks more chunks of the endless greedy loop over all Bm rows and return the newest check the device has completed: finished[64]
// (1 = the row hit EOS / its budget, or is free), n_new[64] (tokens the row holds), *seq_out = number of the chunk that check followed (0:
// none yet), *n_active_out = rows still running then.  Blocks only while more than `lookahead` chunks are unchecked (adaptive, as the batch loop).
extern "C" int sonic_service_step(sonic_engine* e, int n_chunks, int rows, int32_t* finished_out, int32_t* n_new_out, int64_t* seq_out, int32_t* n_active_out) {
    if (!e || n_chunks < 0) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->svc_on) return fail(e, SONIC_ERR_INVALID, "sonic_service_step needs sonic_service_begin");
    const int C = e->opt_decode_chunk > 0 ? e->opt_decode_chunk : 1;
    // rows: the caller's occupied rows all lie below this index (0: all); the chunk runs the next multiple of 16 (graphs captured at
    // sonic_service_begin) - a lightly loaded pool does not pay the step time of a full one (1.15 ms for 1-16 rows, 1.35 for 32, 1.83 for 64)
    int R = rows <= 0 || rows > e->Bm ? e->Bm : (rows + 15) / 16 * 16;
    if (R > e->Bm) R = e->Bm;
    if (rows > 0 && rows <= 2 && e->Bm >= 2 && !e->i8) R = 2;      // one or two sessions' rows: the <= 2-row step (same bits per row as any other row count)
    else if (rows > 0 && rows <= 4 && e->Bm >= 4 && !e->i8 && e->opt_decode_gemv) R = 4;
    auto read_check = [&](bool block) -> int {
        const int i = (int)(e->svc_checked % CHK_RING);
        if (!block) { const hipError_t q = hipEventQuery(e->chk_ev[i]); if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; } if (q != hipSuccess) return -1; }
        else if (hipEventSynchronize(e->chk_ev[i]) != hipSuccess) return -1;
        const int* w = e->svc_h + (size_t)i * SVC_WORDS;
        memcpy(e->svc_fin, w, 64 * 4); memcpy(e->svc_nn, w + 64, 64 * 4); e->svc_active = w[128];
        e->svc_seq = ++e->svc_checked;
        if (e->svc_active < DEV_ERR_ACTIVE) return -2;
        return 1;
    };
    for (int c = 0; c <= n_chunks; ++c) {
        if (c < n_chunks) {
            // rows were running at the last check and the chunk queued last is already complete: the device has been idle waiting for this
            // thread (late by more than a chunk: descheduled, or behind other host work - the Python side of 128 sessions is).  Queue deeper.
            if (e->svc_launched > 0 && e->svc_active > 0 && hipEventQuery(e->chk_ev[(e->svc_launched - 1) % CHK_RING]) == hipSuccess) {
                e->lookahead = e->lookahead * 2 < CHK_MAX_AHEAD ? e->lookahead * 2 : CHK_MAX_AHEAD;
                e->svc_calm = 0; ++e->svc_dry;
            } else if (++e->svc_calm >= 64 && e->lookahead > 1) { e->lookahead /= 2; e->svc_calm = 0; }   // ... and shallower again after a calm stretch: a splice waits
                                                                                                            // behind the queue and finished rows are seen that many chunks late
            (void)hipGetLastError();
            hipGraphExec_t gx = nullptr;
            TRY(chunk_graph(e, R, C, &gx, true));
            HIPC(e, hipGraphLaunch(gx, e->st));
            const int slot = (int)(e->svc_launched % CHK_RING);
            int* w = e->svc_h + (size_t)slot * SVC_WORDS;
            hipLaunchKernelGGL(service_status_kernel, dim3(1), dim3(64), 0, e->st, e->finished, e->n_new, e->n_active, w);
            HIPC(e, hipEventRecord(e->chk_ev[slot], e->st));
            ++e->svc_launched;
        }
        while (e->svc_checked < e->svc_launched) {
            const int r = read_check(e->svc_launched - e->svc_checked > e->lookahead);
            if (r == -2) return fail(e, SONIC_ERR_HIP, "continuous decode loop: a decode kernel gave up on an in-kernel wait, the rows' tokens are invalid");
            if (r < 0) return fail(e, SONIC_ERR_HIP, "continuous decode loop: check event failed: %s", hipGetErrorString(hipGetLastError()));
            if (r == 0) break;
        }
    }
    if (finished_out) memcpy(finished_out, e->svc_fin, 64 * 4);
    if (n_new_out) memcpy(n_new_out, e->svc_nn, 64 * 4);
    if (seq_out) *seq_out = e->svc_seq;
    if (n_active_out) *n_active_out = e->svc_active;
    return SONIC_OK;
}
// the n tokens of a finished row (n from sonic_service_step's n_new), then the row is free for the next splice
extern "C" int sonic_fetch_row(sonic_engine* e, int row, int n, int32_t* out_ids) {
    if (!e || !out_ids) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->svc_on) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_row needs sonic_service_begin");
    if (row < 0 || row >= e->Bm || n < 0 || n > e->out_cap) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_row: row %d / %d tokens out of range", row, n);
    // (a row the newest check saw running is running: releasing it would leave the count of running rows one too high for good)
    if (e->svc_checked > 0 && !e->svc_fin[row]) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_row: row %d has not finished (sonic_service_step's finished[])", row);
    if (n > 0) {
        HIPC(e, hipMemcpyAsync(out_ids, e->out_ids + (size_t)row * e->out_cap, (size_t)n * 4, hipMemcpyDeviceToHost, e->st_io));
        HIPC(e, hipStreamSynchronize(e->st_io));
    }
    hipLaunchKernelGGL(release_row_kernel, dim3(1), dim3(1), 0, e->st, e->kv_len, e->tok_pos, e->finished, row);
    HIPC(e, hipGetLastError());
    return SONIC_OK;
}

// sonic_fetch_row for n rows in one call: the D2H copies go out together, one wait, one release launch - what a block of 32 finished rows costs
// the host drops from 32 lock / copy / wait / launch rounds (about a chunk's worth of time, during which the loop's queue could run dry) to one
__global__ void release_rows_kernel(int* kv_len, int* tok_pos, int* finished, SpliceArgs a, int n) {
    const int i = threadIdx.x;
    if (i < n) { const int row = a.dst[i]; kv_len[row] = 1; tok_pos[row] = 0; finished[row] = 1; }
}
extern "C" int sonic_fetch_rows(sonic_engine* e, int n, const int32_t* rows, const int32_t* counts, int32_t* out_ids, int out_ld) {
    if (!e || !rows || !counts || !out_ids) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->svc_on) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_rows needs sonic_service_begin");
    if (n < 1 || n > 64) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_rows: %d rows", n);
    SpliceArgs a{};
    for (int i = 0; i < n; ++i) {
        const int row = rows[i], c = counts[i];
        if (row < 0 || row >= e->Bm || c < 0 || c > e->out_cap || c > out_ld) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_rows: row %d / %d tokens out of range", row, c);
        if (e->svc_checked > 0 && !e->svc_fin[row]) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_rows: row %d has not finished (sonic_service_step's finished[])", row);
        for (int j = 0; j < i; ++j) if (rows[j] == row) return fail(e, SONIC_ERR_INVALID, "sonic_fetch_rows: row %d named twice", row);
        a.dst[i] = row;
    }
    for (int i = 0; i < n; ++i)
        if (counts[i] > 0) HIPC(e, hipMemcpyAsync(out_ids + (size_t)i * out_ld, e->out_ids + (size_t)rows[i] * e->out_cap, (size_t)counts[i] * 4, hipMemcpyDeviceToHost, e->st_io));
    HIPC(e, hipStreamSynchronize(e->st_io));
    hipLaunchKernelGGL(release_rows_kernel, dim3(1), dim3(64), 0, e->st, e->kv_len, e->tok_pos, e->finished, a, n);
    HIPC(e, hipGetLastError());
    return SONIC_OK;
}

// ---- asynchronous form: the batch runs on a worker thread of the engine's own; the caller's thread returns at once and may drive other slots.
// One job per engine (slot) at a time; between sonic_run_staged_async and sonic_wait the handle takes no other call except ring appends.
static int run_staged_entry(sonic_engine* e) {
    const sonic_engine::AsyncJob& j = e->a_job;
    ENTER(e);
    return run_all(e, j.has_rw ? j.req_win.data() : nullptr, j.R, j.prompt_ids.data(), j.prompt_off.data(), j.max_new.data(), j.want_logits != 0);
}
static void async_loop(sonic_engine* e) {
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(e->a_mu);
            e->a_cv.wait(lk, [&] { return e->a_pending || e->a_stop; });
            if (!e->a_pending) return;                       // (a pending job still runs before the thread leaves)
            e->a_pending = false; e->a_running = true;
        }
        const int st = run_staged_entry(e);
        {
            std::lock_guard<std::mutex> lk(e->a_mu);
            e->a_status = st; e->a_running = false; e->a_done = true;
        }
        e->a_cv.notify_all();
    }
}
static void async_shutdown(sonic_engine* e) {
    {
        std::lock_guard<std::mutex> lk(e->a_mu);
        if (!e->a_started) return;
        e->a_stop = true;
    }
    e->a_cv.notify_all();
    if (e->a_thread.joinable()) e->a_thread.join();
}
extern "C" int sonic_run_staged_async(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                                      const int32_t* max_new, int want_step_logits) {
    if (!e || !prompt_ids || !prompt_off || !max_new || R < 1 || R > 64) return SONIC_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->a_mu);
    if (e->a_stop) return SONIC_ERR_INVALID;
    if (e->a_pending || e->a_running || e->a_done) {         // (e->err belongs to the engine lock; report through the creating thread's slot)
        return fail(nullptr, SONIC_ERR_INVALID, "sonic_run_staged_async: the previous asynchronous run of this handle has not been waited for (sonic_wait)");
    }
    sonic_engine::AsyncJob& j = e->a_job;
    j.R = R; j.want_logits = want_step_logits; j.has_rw = req_win != nullptr;
    j.req_win.assign(req_win ? req_win : nullptr, req_win ? req_win + R + 1 : nullptr);
    j.prompt_off.assign(prompt_off, prompt_off + R + 1);
    if (prompt_off[0] != 0 || prompt_off[R] < prompt_off[0]) return fail(nullptr, SONIC_ERR_INVALID, "sonic_run_staged_async: prompt_off must start at 0 and ascend");
    j.prompt_ids.assign(prompt_ids, prompt_ids + prompt_off[R]);
    j.max_new.assign(max_new, max_new + R);
    if (!e->a_started) { e->a_thread = std::thread(async_loop, e); e->a_started = true; }
    e->a_pending = true;
    e->a_cv.notify_all();
    return SONIC_OK;
}
// blocks until the asynchronous run of this handle is complete and returns ITS status (sonic_last_error(e) has the text); SONIC_OK at once when
// nothing is outstanding.  *busy_out (optional, with block == 0): 1 while the run is still going, and the call returns SONIC_OK without waiting.
extern "C" int sonic_wait(sonic_engine* e, int block, int32_t* busy_out) {
    if (!e) return SONIC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(e->a_mu);
    if (busy_out) *busy_out = 0;
    if (!e->a_pending && !e->a_running && !e->a_done) return SONIC_OK;
    if (!block && !e->a_done) { if (busy_out) *busy_out = 1; return SONIC_OK; }
    e->a_cv.wait(lk, [&] { return e->a_done; });
    e->a_done = false;
    return e->a_status;
}

// Stage entry points: the two halves of sonic_run_staged.  sonic_prefill = log-mel, encoder, projector, decoder prefill
// and the first greedy token of every request (generate()'s first forward, HF:generation/utils.py:2612-2616); sonic_decode_step = up to
// n_steps further iterations of the greedy loop (:2876-2943); *n_active_out = rows that are neither at EOS nor at their budget,
// *steps_done_out = steps actually run (fewer than asked once the largest budget is reached).  sonic_fetch_tokens reads the result at any point.
extern "C" int sonic_prefill(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                             const int32_t* max_new, int want_step_logits) {
    if (!e || !prompt_ids || !prompt_off || !max_new) return SONIC_ERR_INVALID;
    ENTER(e);
    TRY(run_to_first_token(e, req_win, R, prompt_ids, prompt_off, max_new, want_step_logits != 0));
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    return SONIC_OK;
}
// sonic_prefill without the closing wait: everything up to the first token is QUEUED on the handle's stream when the call returns.  For the
// pipeline form: sonic_splice_rows orders its copies behind this work on the device, so the host need not come back between prefill and splice.
extern "C" int sonic_prefill_enqueue(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off, const int32_t* max_new) {
    if (!e || !prompt_ids || !prompt_off || !max_new) return SONIC_ERR_INVALID;
    ENTER(e);
    TRY(run_to_first_token(e, req_win, R, prompt_ids, prompt_off, max_new, false));
    HIPC(e, hipGetLastError());
    return SONIC_OK;
}
extern "C" int sonic_decode_step(sonic_engine* e, int n_steps, int32_t* n_active_out, int32_t* steps_done_out) {
    if (!e || n_steps < 0) return SONIC_ERR_INVALID;
    ENTER(e);
    if (e->R < 1 || e->greedy_calls < 1) return fail(e, SONIC_ERR_INVALID, "sonic_decode_step needs a batch that went through sonic_prefill");
    int done = 0;
    TRY(run_decode_steps(e, n_steps, &done));
    HIPC(e, hipMemcpyAsync(e->n_active_h + CHK_RING, e->n_active, 4, hipMemcpyDeviceToHost, e->st));
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    if (n_active_out) *n_active_out = e->n_active_h[CHK_RING];
    if (steps_done_out) *steps_done_out = done;
    return SONIC_OK;
}

static int fetch_locked(sonic_engine* e, int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits) {
    const int R = e->R;
    if (R < 1) return fail(e, SONIC_ERR_INVALID, "nothing to fetch");
    std::vector<int> nn(64), kvl(64), tps(64), fin(64);
    int act_err[2] = {0, 0};
    HIPC(e, d2h_async(e, act_err, e->n_active, 2 * 4));
    HIPC(e, d2h_async(e, nn.data(), e->n_new, 64 * 4));
    HIPC(e, d2h_async(e, kvl.data(), e->kv_len, 64 * 4));
    HIPC(e, d2h_async(e, tps.data(), e->tok_pos, 64 * 4));
    HIPC(e, d2h_async(e, fin.data(), e->finished, 64 * 4));
    HIPC(e, stream_sync(e));
    if (act_err[1] != 0) return fail(e, SONIC_ERR_HIP, "a decode kernel gave up on an in-kernel wait (device error word %d): the batch's tokens are invalid", act_err[1]);
    // invariants of the greedy controller: a running row's context grows by one per launch; a finished row stopped growing with the
    // launch that finished it (kv_len = prompt + tokens - 1), so no row ever leaves its [max_ctx] cache region
    for (int r = 0; r < R && r < (int)e->last_qlen.size(); ++r) {
        const int want_kv = e->last_qlen[r] + nn[r] - (fin[r] ? 1 : 0);
        const int want_new = e->greedy_calls < e->last_maxnew[r] ? e->greedy_calls : e->last_maxnew[r];
        const bool pos_ok = nn[r] >= 1 && kvl[r] == want_kv && kvl[r] <= e->max_ctx && (tps[r] == kvl[r] - 1 || (fin[r] && nn[r] == 1));
        if (!pos_ok || nn[r] > e->last_maxnew[r] || (!fin[r] && nn[r] != e->greedy_calls) || (e->d.n_eos == 0 && nn[r] != want_new))
            return fail(e, SONIC_ERR_HIP, "decoder state check failed for request %d: kv_len %d (expected %d), tok_pos %d, n_new %d, finished %d (budget %d, greedy launches %d)",
                        r, kvl[r], want_kv, tps[r], nn[r], fin[r], e->last_maxnew[r], e->greedy_calls);
    }
    for (int r = 0; r < R; ++r) {
        if (out_len) out_len[r] = nn[r];
        if (out_ids) {
            if (nn[r] > out_ld) return fail(e, SONIC_ERR_INVALID, "out_ld too small");
            HIPC(e, d2h_async(e, out_ids + (size_t)r * out_ld, e->out_ids + (size_t)r * e->out_cap, (size_t)nn[r] * 4));
        }
    }
    if (step_logits) {
        if (!e->dump_steps) return fail(e, SONIC_ERR_INVALID, "step logits were not requested for the last run");
        HIPC(e, d2h_async(e, step_logits, e->dump, (size_t)e->dump_steps * R * e->d.vocab * 4));
    }
    HIPC(e, stream_sync(e));
    return SONIC_OK;
}

extern "C" int sonic_fetch_tokens(sonic_engine* e, int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    return fetch_locked(e, out_ids, out_ld, out_len, step_logits);
}

extern "C" int sonic_transcribe_batch(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int W, const int32_t* req_win, int R,
                                      const int32_t* prompt_ids, const int64_t* prompt_off, const int32_t* max_new,
                                      int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits) {
    if (!e || !prompt_ids || !prompt_off || !max_new) return SONIC_ERR_INVALID;
    ENTER(e);
    TRY(stage_pcm_locked(e, pcm, offsets, W));
    TRY(run_all(e, req_win, R, prompt_ids, prompt_off, max_new, step_logits != nullptr));
    return fetch_locked(e, out_ids, out_ld, out_len, step_logits);
}

// Teacher forcing for parity tests (oracle_outputs.force_ids, sonic_oracle.c): while set, every run feeds ids[r][n] as token n of
// request r instead of the argmax (logits are still computed and dumped; EOS / budget rules apply to the forced token).  NULL clears.
extern "C" int sonic_set_forced_ids(sonic_engine* e, const int32_t* ids, int R, int ld) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    HIPC(e, stream_sync(e));
    if (e->force_d) { (void)hipFree(e->force_d); e->force_d = nullptr; e->force_ld = e->force_R = 0; }
    if (!ids) return SONIC_OK;
    if (R < 1 || R > e->Bm || ld < 1) return fail(e, SONIC_ERR_INVALID, "forced ids: bad shape [%d][%d]", R, ld);
    for (long i = 0; i < (long)R * ld; ++i)
        if (ids[i] < 0 || ids[i] >= e->d.vocab) return fail(e, SONIC_ERR_INVALID, "forced id %d out of vocabulary", ids[i]);
    HIPC(e, hipMalloc((void**)&e->force_d, (size_t)R * ld * 4));
    HIPC(e, h2d(e, e->force_d, ids, (size_t)R * ld * 4));
    e->force_R = R; e->force_ld = ld;
    return SONIC_OK;
}

extern "C" int sonic_get_timings(sonic_engine* e, sonic_timings* out) {
    if (!e || !out) return SONIC_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    *out = e->tim;
    return SONIC_OK;
}

// ------------------------------------------------------------------------------------------ C ABI: stage entry points
extern "C" int sonic_logmel(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int B, float* feats_out, int32_t* mask_out) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    TRY(stage_pcm_locked(e, pcm, offsets, B));
    TRY(run_mel(e, B, feats_out != nullptr));
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    const sonic_dims& d = e->d;
    if (feats_out) HIPC(e, d2h(e, feats_out, e->feats_f32, (size_t)B * d.n_mels * d.n_frames * 4));
    if (mask_out)
        for (int b = 0; b < B; ++b) {
            const int v = frames_of(e->n_samples_h[b]);
            for (int t = 0; t < d.n_frames; ++t) mask_out[(size_t)b * d.n_frames + t] = t < v ? 1 : 0;   // attention_mask[:, ::160]
        }
    return SONIC_OK;
}

extern "C" int sonic_encode(sonic_engine* e, const float* feats, const int32_t* n_valid_frames, int B,
                            float* embeds_out, int32_t* n_audio_out, float* enc_layers_out, float* enc_out) {
    if (!e || !feats || !n_valid_frames) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->finalized) return fail(e, SONIC_ERR_INVALID, "weights not finalized");
    if (B < 1 || B > e->Bm) return fail(e, SONIC_ERR_INVALID, "batch out of range");
    const sonic_dims& d = e->d;
    const size_t n = (size_t)B * d.n_mels * d.n_frames;
    if (e->f32) {
        if (!e->feats_f32) HIPC(e, hipMalloc((void**)&e->feats_f32, (size_t)e->Bm * d.n_mels * d.n_frames * 4));
        HIPC(e, h2d(e, e->feats_f32, feats, n * 4));
        TRY(f32_run_encoder(e, B, enc_layers_out, enc_out));
        HIPC(e, stream_sync(e)); HIPC(e, hipGetLastError());
        if (embeds_out) HIPC(e, d2h(e, embeds_out, e->f->pe, (size_t)B * e->Ta * d.dec_d * 4));
        if (n_audio_out) for (int b = 0; b < B; ++b) n_audio_out[b] = keep_rows(d, n_valid_frames[b]);
        return SONIC_OK;
    }
    float* tmp = nullptr;
    HIPC(e, hipMalloc((void**)&tmp, n * 4));
    hipError_t r = h2d(e, tmp, feats, n * 4);
    if (r != hipSuccess) { (void)hipFree(tmp); HIPC(e, r); }
    const long per = (long)d.n_mels * d.n_frames;
    DT_SWITCH(e->dt, T, hipLaunchKernelGGL(feats_to_fm_kernel<T>, dim3((per + 255) / 256, B), dim3(256), 0, e->st, tmp, (T*)e->feats_fm, d.n_mels, d.n_frames));
    if (e->i8) HIPC(e, hipMemcpyAsync(e->win_req, e->seq_iota, (size_t)B * 4, hipMemcpyDeviceToDevice, e->st));   // every window its own request
    int s = run_encoder(e, B, enc_layers_out, enc_out, B);
    hipError_t r2 = stream_sync(e);
    (void)hipFree(tmp);
    TRY(s); HIPC(e, r2); HIPC(e, hipGetLastError());
    if (embeds_out) {
        float* t2 = nullptr;
        const size_t m = (size_t)B * e->Ta * d.dec_d;
        HIPC(e, hipMalloc((void**)&t2, m * 4));
        launch_bf16_to_f32(e->pe, t2, (long)m, e->st, e->dt);
        hipError_t r3 = stream_sync(e);
        if (r3 == hipSuccess) r3 = d2h(e, embeds_out, t2, m * 4);
        (void)hipFree(t2);
        HIPC(e, r3);
    }
    if (n_audio_out) for (int b = 0; b < B; ++b) n_audio_out[b] = keep_rows(d, n_valid_frames[b]);
    return SONIC_OK;
}

// ------------------------------------------------------------------------------------------ C ABI: kernel test hooks
struct TmpBuf {
    std::vector<void*> v;
    hipStream_t st;
    explicit TmpBuf(hipStream_t s) : st(s) {}
    ~TmpBuf() { for (void* p : v) (void)hipFree(p); }
    // Zero-fill with a KERNEL on the engine stream.  hipMemsetAsync on this non-blocking stream was seen not to be reliably ordered
    // against its neighbours (a stale log-mel maximum survived one in round 1); here a late zero fill would wipe a buffer that a
    // conversion kernel or a GEMM has already written - the signature of the intermittent test_gemm256_path failures (gross errors on
    // a few tiles, clean on an immediate rerun, never in the engine's own long-lived buffers).
    template <typename Tt> Tt* get(size_t n) {
        void* p = nullptr;
        const size_t bytes = (((n ? n : 1) * sizeof(Tt)) + 3) / 4 * 4;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        size_t left = bytes / 4; int* q = (int*)p;
        while (left > 0) { const int c = left > (1u << 30) ? (1 << 30) : (int)left; launch_fill_i32(q, 0, c, st); q += c; left -= c; }
        (void)hipStreamSynchronize(st);
        v.push_back(p);
        return (Tt*)p;
    }
};
static bf16_t* up_bf16(sonic_engine* e, TmpBuf& tb, const float* h, size_t n, size_t pad = 0) {
    float* f = tb.get<float>(n); bf16_t* b = tb.get<bf16_t>(n + pad);
    if (!f || !b) return nullptr;
    if (h2d(e, f, h, n * 4) != hipSuccess) return nullptr;
    launch_f32_to_bf16(f, b, (long)n, e->st, e->dt);     // the engine's element type: bf16, or fp16 on an int8-mode engine
    return b;
}
static float* up_f32(sonic_engine* e, TmpBuf& tb, const float* h, size_t n) {
    float* f = tb.get<float>(n);
    if (f && h2d(e, f, h, n * 4) != hipSuccess) return nullptr;
    return f;
}
static int down_bf16(sonic_engine* e, TmpBuf& tb, const bf16_t* d, float* h, size_t n) {
    float* f = tb.get<float>(n);
    if (!f) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    launch_bf16_to_f32(d, f, (long)n, e->st, e->dt);
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    HIPC(e, d2h(e, h, f, n * 4));
    return SONIC_OK;
}

extern "C" int sonic_test_gemm(sonic_engine* e, const float* A, const float* W, const float* bias, const float* resid, float* C,
                               int M, int N, int K, int epi) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (K % 64 || N % 4) return fail(e, SONIC_ERR_INVALID, "K must be a multiple of 64 and N of 4");
    TmpBuf tb(e->st);
    const int Nout = (epi == EPI_SWIGLU) ? N / 2 : N;
    bf16_t* dA = up_bf16(e, tb, A, (size_t)M * K); bf16_t* dW = up_bf16(e, tb, W, (size_t)N * K);
    float* db = bias ? up_f32(e, tb, bias, N) : nullptr;
    bf16_t* dR = resid ? up_bf16(e, tb, resid, (size_t)M * Nout) : nullptr;
    bf16_t* dC = tb.get<bf16_t>((size_t)M * Nout);
    if (!dA || !dW || !dC) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    gemm(e, epi, dA, K, dW, db, dC, Nout, M, N, K, dR, Nout);
    return down_bf16(e, tb, dC, C, (size_t)M * Nout);
}

extern "C" int sonic_test_skinny(sonic_engine* e, const float* X, const float* W, float* C, int M, int N, int K) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (M < 1 || M > 64 || N % 16 || K % 256 || skinny_pick_ksplit(N, K) < 1) return fail(e, SONIC_ERR_INVALID, "skinny: M<=64, N%%16==0, K%%256==0");
    TmpBuf tb(e->st);
    bf16_t* dX = up_bf16(e, tb, X, (size_t)M * K); bf16_t* dW = up_bf16(e, tb, W, (size_t)N * K);
    bf16_t* dWt = tb.get<bf16_t>((size_t)N * K);
    const int ks = skinny_pick_ksplit(N, K), mpad = ((M + 15) / 16) * 16;
    float* P = tb.get<float>((size_t)ks * mpad * N);
    if (!dX || !dW || !dWt || !P) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    launch_tile_weights(dW, dWt, N, K, e->st);
    SkinnyArgs a{}; a.X = dX; a.ldx = K; a.W = dWt; a.P = P; a.M = M; a.N = N; a.K = K; a.ksplit = ks; a.dt = e->dt;
    launch_skinny(a, e->st);
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    std::vector<float> h((size_t)ks * mpad * N);
    HIPC(e, d2h(e, h.data(), P, h.size() * 4));
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float s = 0;
            for (int k = 0; k < ks; ++k) s += h[((size_t)k * mpad + m) * N + n];
            C[(size_t)m * N + n] = s;
        }
    return SONIC_OK;
}

__global__ void test_transpose_v_kernel(const bf16_t* v, bf16_t* vt, int B, int Tk, int Hkv, int hd, int ld_t) {
    // v [B][Tk][Hkv*hd] -> vt [B][Hkv][hd][ld_t]
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * Tk * Hkv * hd) return;
    const int c = e % (Hkv * hd), t = (e / (Hkv * hd)) % Tk, b = e / ((long)Hkv * hd * Tk);
    vt[(((long)b * Hkv + c / hd) * hd + c % hd) * ld_t + t] = v[e];
}

extern "C" int sonic_test_attention(sonic_engine* e, const float* q, const float* k, const float* v, float* out,
                                    int B, int Tq, int Tk, int Hq, int Hkv, int hd, int causal) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (hd != 64 && hd != 128) return fail(e, SONIC_ERR_INVALID, "hd must be 64 or 128");
    TmpBuf tb(e->st);
    const int Tkp = (Tk + 63) / 64 * 64;
    bf16_t* dq = up_bf16(e, tb, q, (size_t)B * Tq * Hq * hd);
    bf16_t* dk = up_bf16(e, tb, k, (size_t)B * Tk * Hkv * hd, (size_t)64 * Hkv * hd);
    bf16_t* dv = up_bf16(e, tb, v, (size_t)B * Tk * Hkv * hd);
    bf16_t* dvt = tb.get<bf16_t>((size_t)B * Hkv * hd * Tkp);
    bf16_t* dO = tb.get<bf16_t>((size_t)B * Tq * Hq * hd);
    if (!dq || !dk || !dv || !dvt || !dO) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    const long nv = (long)B * Tk * Hkv * hd;
    hipLaunchKernelGGL(test_transpose_v_kernel, dim3((nv + 255) / 256), dim3(256), 0, e->st, dv, dvt, B, Tk, Hkv, hd, Tkp);
    FlashArgs f{};
    f.Q = dq; f.q_ld = (long)Hq * hd; f.K = dk; f.k_ld = (long)Hkv * hd; f.Vt = dvt; f.vt_ld = Tkp; f.O = dO; f.o_ld = (long)Hq * hd;
    f.q_seq_stride = (long)Tq * Hq * hd; f.k_seq_stride = (long)Tk * Hkv * hd; f.k_head_stride = hd;
    f.vt_seq_stride = (long)Hkv * hd * Tkp; f.vt_head_stride = (long)hd * Tkp; f.T = Tq; f.Hq = Hq; f.Hkv = Hkv; f.scale = 1.0f / sqrtf((float)hd); f.dt = e->dt;
    int *ql = nullptr, *kl = nullptr;
    if (Tq != Tk) {   // per-sequence lengths (decode-style offset: query t sits at position Tk - Tq + t)
        ql = tb.get<int>(B); kl = tb.get<int>(B);
        std::vector<int> a(B, Tq), b2(B, Tk);
        HIPC(e, h2d(e, ql, a.data(), B * 4)); HIPC(e, h2d(e, kl, b2.data(), B * 4));
        f.q_len = ql; f.kv_len = kl;
    }
    launch_flash(f, hd, causal != 0, B, Tq, e->st);
    return down_bf16(e, tb, dO, out, (size_t)B * Tq * Hq * hd);
}

extern "C" int sonic_test_decode_attention(sonic_engine* e, const float* q, const float* k, const float* v, float* out, int B, int Tk, int Hq, int Hkv) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    const int hd = 128, ctx = (Tk + 63) / 64 * 64;
    if (Hq % Hkv || Hq / Hkv > 4) return fail(e, SONIC_ERR_INVALID, "bad GQA group");
    TmpBuf tb(e->st);
    // k, v given as [B][Tk][Hkv*hd]; cache layout is [B][Hkv][ctx][hd]
    std::vector<float> kc((size_t)B * Hkv * ctx * hd, 0.f), vc(kc.size(), 0.f);
    for (int b = 0; b < B; ++b) for (int t = 0; t < Tk; ++t) for (int h = 0; h < Hkv; ++h) for (int i = 0; i < hd; ++i) {
        const size_t s = (((size_t)b * Tk + t) * Hkv + h) * hd + i, dd = (((size_t)b * Hkv + h) * ctx + t) * hd + i;
        kc[dd] = k[s]; vc[dd] = v[s];
    }
    bf16_t* dq = up_bf16(e, tb, q, (size_t)B * Hq * hd); bf16_t* dk = up_bf16(e, tb, kc.data(), kc.size()); bf16_t* dv = up_bf16(e, tb, vc.data(), vc.size());
    bf16_t* dO = tb.get<bf16_t>((size_t)B * Hq * hd); int* kl = tb.get<int>(B);
    if (!dq || !dk || !dv || !dO || !kl) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    std::vector<int> l(B, Tk); HIPC(e, h2d(e, kl, l.data(), B * 4));
    DecodeAttnArgs a{}; a.Q = dq; a.P = nullptr; a.Kc = dk; a.Vc = dv; a.O = dO; a.kv_len = kl; a.Hq = Hq; a.Hkv = Hkv; a.ctx_max = ctx; a.scale = 1.0f / sqrtf(128.f); a.dt = e->dt;
    launch_decode_attn(a, B, e->st);
    return down_bf16(e, tb, dO, out, (size_t)B * Hq * hd);
}

extern "C" int sonic_test_layernorm(sonic_engine* e, const float* x, const float* w, const float* b, float* y, int rows, int d, float eps, int rms) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (d % 8 || d > 2048) return fail(e, SONIC_ERR_INVALID, "d must be a multiple of 8 and <= 2048");
    TmpBuf tb(e->st);
    bf16_t* dx = up_bf16(e, tb, x, (size_t)rows * d); float* dw = up_f32(e, tb, w, d); float* db = b ? up_f32(e, tb, b, d) : nullptr;
    bf16_t* dy = tb.get<bf16_t>((size_t)rows * d);
    if (!dx || !dw || !dy) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    if (rms) launch_rmsnorm(dx, dw, dy, rows, d, eps, nullptr, e->st, e->dt);
    else launch_layernorm(dx, dw, db, dy, rows, d, eps, e->st, e->dt);
    return down_bf16(e, tb, dy, y, (size_t)rows * d);
}

extern "C" int sonic_bench_gemm(sonic_engine* e, int M, int N, int K, int epi, int iters, float* ms_per_launch) {
    if (!e || !ms_per_launch) return SONIC_ERR_INVALID;
    ENTER(e);
    if (K % 64 || N % 4 || iters < 1) return fail(e, SONIC_ERR_INVALID, "bad gemm bench shape");
    TmpBuf tb(e->st);
    const int Nout = (epi == EPI_SWIGLU) ? N / 2 : N;
    bf16_t* dA = tb.get<bf16_t>((size_t)M * K + 1024); bf16_t* dW = tb.get<bf16_t>((size_t)N * K); bf16_t* dC = tb.get<bf16_t>((size_t)M * Nout);
    float* db = tb.get<float>(N);
    bf16_t* dVt = nullptr;
    GemmArgs a{};
    if (epi == EPI_QKV_VT) {   // encoder QKV shape: last third of the columns is V, written transposed per 1500-frame segment
        if (N % 3 || M % 1500) return fail(e, SONIC_ERR_INVALID, "QKV bench needs N % 3 == 0 and M % 1500 == 0");
        dVt = tb.get<bf16_t>((size_t)(M / 1500) * (N / 3) * 1536);
        a.Vt = dVt; a.n_split = 2 * N / 3; a.seg_T = 1500; a.vt_ld = 1536; a.vt_seg_stride = (long)(N / 3) * 1536;
    }
    if (!dA || !dW || !dC || !db || (epi == EPI_QKV_VT && !dVt)) return fail(e, SONIC_ERR_OOM, "HIP out of memory in gemm bench");
    // random (not zero) operands: zero data reads high on this chip (cdna_hip_programming.md rule 25)
    launch_synth_fill(0x1234, (long)M * K, 1.0f, 0.f, dA, nullptr, e->st);
    launch_synth_fill(0x5678, (long)N * K, 0.05f, 0.f, dW, nullptr, e->st);
    a.A = dA; a.lda = K; a.W = dW; a.C = dC; a.ldc = (epi == EPI_QKV_VT) ? 2 * N / 3 : Nout; a.bias = db; a.R = dC; a.ldr = Nout; a.M = M; a.N = N; a.K = K; a.batch = 1; a.dt = e->dt;
    a.gelu_lut = e->opt_no_gelu_lut ? nullptr : e->gelu_lut;          // as the encoder's fc1 (round 5: the bench used to time the arithmetic GELU)
    for (int i = 0; i < 2; ++i) launch_gemm(a, epi, e->st);
    if (e->opt_gemm_trace) {
        // diagnostics: where a 256x256 tile's time goes (in-kernel 100 MHz stamps of every block of ONE launch) and how long a CU waits between two blocks
        const int nblk = ((M + 255) / 256) * ((N + 255) / 256);
        long long* dbg = tb.get<long long>((size_t)nblk * 8);
        if (dbg) {
            (void)hipMemsetAsync(dbg, 0, (size_t)nblk * 64, e->st);
            GemmArgs t = a; t.dbg = dbg;
            launch_gemm(t, epi, e->st);
            std::vector<long long> h((size_t)nblk * 8);
            if (d2h(e, h.data(), dbg, (size_t)nblk * 64) == hipSuccess) {
                std::map<long long, std::vector<std::pair<long long, long long>>> per_cu;     // hw id -> (entry, exit)
                double s01 = 0, s12 = 0, s23 = 0; int n = 0;
                for (int b = 0; b < nblk; ++b) {
                    const long long* r = &h[(size_t)b * 8];
                    if (!r[0] || !r[3]) continue;
                    s01 += (r[1] - r[0]) * 0.01; s12 += (r[2] - r[1]) * 0.01; s23 += (r[3] - r[2]) * 0.01; ++n;
                    per_cu[r[4] & 0x0000000F0000FF00ll].push_back({r[0], r[3]});            // XCC_ID[3:0] | HW_ID: se_id[15:13] sh_id[12] cu_id[11:8]
                }
                double gap = 0; int ng = 0; long long t_first = 0, t_last = 0;
                for (auto& kv : per_cu) {
                    auto& v = kv.second; std::sort(v.begin(), v.end());
                    for (size_t i = 1; i < v.size(); ++i) { gap += (v[i].first - v[i - 1].second) * 0.01; ++ng; }
                    for (auto& x : v) { if (!t_first || x.first < t_first) t_first = x.first; if (x.second > t_last) t_last = x.second; }
                }
                fprintf(stderr, "[gemm_trace] M=%d N=%d K=%d epi=%d: %d blocks on %zu CUs; per block: entry -> first K tile landed %.2f us, K loop %.2f us, epilogue %.2f us; "
                                "gap between consecutive blocks of a CU %.2f us (n=%d); first entry -> last exit %.1f us\n",
                        M, N, K, epi, n, per_cu.size(), s01 / n, s12 / n, s23 / n, ng ? gap / ng : 0.0, ng, (t_last - t_first) * 0.01);
            }
        }
    }
    hipEvent_t ea, eb; HIPC(e, hipEventCreate(&ea)); HIPC(e, hipEventCreate(&eb));
    (void)hipEventRecord(ea, e->st);
    for (int i = 0; i < iters; ++i) launch_gemm(a, epi, e->st);
    (void)hipEventRecord(eb, e->st);
    hipError_t r = stream_sync(e);
    float ms = 0; (void)hipEventElapsedTime(&ms, ea, eb);
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    HIPC(e, r); HIPC(e, hipGetLastError());
    *ms_per_launch = ms / iters;
    return SONIC_OK;
}

extern "C" int sonic_bench_skinny(sonic_engine* e, int M, int N, int K, int variant, int iters, float* us_per_launch) {
    if (!e || !us_per_launch) return SONIC_ERR_INVALID;
    ENTER(e);
    if (M < 1 || M > 64 || N % 16 || K % 256 || skinny_pick_ksplit(N, K) < 1 || iters < 1) return fail(e, SONIC_ERR_INVALID, "bad skinny bench shape");
    TmpBuf tb(e->st);
    g_opts.skinny_variant = variant;   // this call only (ENTER() reloads the engine's own knobs on the next entry); before the ksplit pick: the slab count depends on the kernel family
    // 8 distinct weight copies so consecutive launches do not re-read an Infinity-Cache-resident matrix
    const int copies = 8;
    bf16_t* dW = tb.get<bf16_t>((size_t)copies * N * K); bf16_t* dX = tb.get<bf16_t>((size_t)64 * K);
    const int ks = skinny_pick_ksplit(N, K), mpad = ((M + 15) / 16) * 16;
    float* P = tb.get<float>((size_t)ks * mpad * N);
    if (!dW || !dX || !P) return fail(e, SONIC_ERR_OOM, "HIP out of memory in skinny bench");
    launch_synth_fill(0x77, (long)copies * N * K, 0.05f, 0.f, dW, nullptr, e->st);
    launch_synth_fill(0x78, (long)64 * K, 1.0f, 0.f, dX, nullptr, e->st);
    SkinnyArgs a{}; a.X = dX; a.ldx = K; a.P = P; a.M = M; a.N = N; a.K = K; a.ksplit = ks; a.dt = e->dt;
    for (int i = 0; i < copies; ++i) { a.W = dW + (size_t)(i % copies) * N * K; launch_skinny(a, e->st); }
    hipEvent_t ea, eb; HIPC(e, hipEventCreate(&ea)); HIPC(e, hipEventCreate(&eb));
    (void)hipEventRecord(ea, e->st);
    for (int i = 0; i < iters; ++i) { a.W = dW + (size_t)(i % copies) * N * K; launch_skinny(a, e->st); }
    (void)hipEventRecord(eb, e->st);
    hipError_t r = stream_sync(e);
    float ms = 0; (void)hipEventElapsedTime(&ms, ea, eb);
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    g_opts = e->opts;
    HIPC(e, r); HIPC(e, hipGetLastError());
    *us_per_launch = ms * 1e3f / iters;
    return SONIC_OK;
}
static void drop_graphs(sonic_engine* e) { for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.second); e->graphs.clear(); }
extern "C" int sonic_set_option(sonic_engine* e, const char* key, int value) {
    if (!e || !key) return SONIC_ERR_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    // knobs live in the engine: two engines in one process do not see each other's settings; captured decode graphs of THIS engine
    // are dropped whenever a knob that changes the captured kernels moves
    if (!strcmp(key, "skinny_variant")) { e->opts.skinny_variant = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "gemm_force128")) { e->opts.gemm_force128 = value; return SONIC_OK; }
    if (!strcmp(key, "no_fused_gu")) { e->opts.no_fused_gu = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "no_fused_gu64")) { e->opts.no_fused_gu64 = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "gu64_two_pass")) { e->opts.gu64_two_pass = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "gu64_split_norm")) { e->opts.gu64_split_norm = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "ktrace_wave")) { e->opts.ktrace_wave = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "no_skinny768")) { e->opts.no_skinny768 = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "no_skinny48")) { e->opts.no_skinny48 = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "o64_16rows")) { e->opts.o64_16rows = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "i8_no_lnq")) { e->opt_i8_no_lnq = value; return SONIC_OK; }      // int8 encoder: LayerNorm does not quantise its rows (A/B)
    if (!strcmp(key, "i8_no_qkv_fuse")) { e->opt_i8_no_qkv_fuse = value; return SONIC_OK; }   // int8 encoder: RoPE and V^T as their own passes (A/B)
    if (!strcmp(key, "i8_dbg")) { e->opt_i8_dbg = value; drop_graphs(e); return SONIC_OK; }     // timing experiments (wrong results)
    if (!strcmp(key, "i8_no_xq")) { e->opt_i8_no_xq = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "gemm_small_eff")) { e->opts.gemm_small_eff = value; return SONIC_OK; }
    if (!strcmp(key, "gemm128_shallow")) { e->opts.gemm128_shallow = value; return SONIC_OK; }
    if (!strcmp(key, "no_skinny_i8_wide")) { e->opts.no_skinny_i8_wide = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "gemm256_stagger")) { e->opts.gemm256_stagger = value; return SONIC_OK; }
    if (!strcmp(key, "flash_variant")) { e->opts.flash_variant = value; return SONIC_OK; }
    if (!strcmp(key, "flash_enc")) { e->opts.flash_enc = value; return SONIC_OK; }          // 0: rounds 1-4's encoder attention; v > 0: flash_enc_kernel mode v - 1
    if (!strcmp(key, "gemm256_persist")) { e->opts.gemm256_persist = value; return SONIC_OK; }
    if (!strcmp(key, "gemm256_persist_cus")) { e->opts.gemm256_persist_cus = value > 0 ? value : 256; return SONIC_OK; }
    if (!strcmp(key, "gemm256_gm")) { e->opts.gemm256_gm = value > 0 ? value : 8; return SONIC_OK; }   // raster group height of the 256x256 GEMM (experiments)
    if (!strcmp(key, "i8_defer_thr")) { e->opt_i8_defer_thr = value; return SONIC_OK; }   // int8: outlier lists longer than this go to the dense side product (-1: never)
    if (!strcmp(key, "decode_prefetch")) { e->opts.decode_prefetch = value; drop_graphs(e); return SONIC_OK; }   // idle-CU weight prefetch (experiment)
    if (!strcmp(key, "decode_attn_occ2")) { e->opts.decode_attn_occ2 = value; drop_graphs(e); return SONIC_OK; }   // decode attention at 128 VGPRs (two blocks per CU can co-reside; A/B)
    if (!strcmp(key, "decode_attn_v1")) { e->opts.decode_attn_v1 = value; drop_graphs(e); return SONIC_OK; }   // round 2's VALU P.V decode attention (A/B)
    if (!strcmp(key, "prefill_taps")) { e->taps_on = value; return SONIC_OK; }
    if (!strcmp(key, "no_pre_norm")) { e->opt_no_pre_norm = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "decode_gemv")) { e->opt_decode_gemv = value; drop_graphs(e); return SONIC_OK; }
    if (!strcmp(key, "f32_synth_bf16")) { e->opt_f32_synth_bf16 = value; return SONIC_OK; }
    if (!strcmp(key, "no_pre_norm")) { e->opt_no_pre_norm = value; drop_graphs(e); return SONIC_OK; }   // <= 2 rows: standalone add+RMSNorm launches as for more rows (A/B, same bits)
    if (!strcmp(key, "no_graph")) { e->opt_no_graph = value; return SONIC_OK; }            // eager decode loop (debugging)
    if (!strcmp(key, "decode_lookahead")) { e->lookahead = value < 1 ? 1 : (value > CHK_MAX_AHEAD ? CHK_MAX_AHEAD : value); return SONIC_OK; }   // start value (it adapts)
    if (!strcmp(key, "decode_chunk")) { e->opt_decode_chunk = value > 0 ? (value > 64 ? 64 : value) : 1; return SONIC_OK; }   // token steps per graph launch / early-stop check
    if (!strcmp(key, "prefill_rowmajor")) { e->opt_prefill_rowmajor = value; return SONIC_OK; } // prefill GEMMs read the row-major decoder weights (kept only under SONIC_KEEP_ROWMAJOR=1; A/B)
    if (!strcmp(key, "no_rope_tiles")) { e->opt_no_rope_tiles = value; return SONIC_OK; }  // prefill RoPE + KV append per token (rounds 1-4) instead of per 16-position tile (A/B)
    if (!strcmp(key, "gemm_trace")) { e->opt_gemm_trace = value; return SONIC_OK; }        // sonic_bench_gemm prints an in-kernel timeline of one launch to stderr
    if (!strcmp(key, "gemm_timing")) { e->opt_gemm_timing = value; return SONIC_OK; }      // HIP events around every encoder-layer GEMM launch
    if (!strcmp(key, "no_fused_rope")) { e->opt_no_fused_rope = value; return SONIC_OK; }  // encoder RoPE as its own pass (A/B against the fused epilogue)
    if (!strcmp(key, "ktrace")) {              // diagnostics: record in-kernel timestamps of decoder layer `value` (-1: off); sonic_debug_ktrace reads them
        HIPC(e, hipSetDevice(e->device));
        if (value >= 0 && !e->kt) { TRY(dalloc(e, &e->kt, (size_t)8 * KT_SLOT_BLOCKS * 8)); }
        if (e->kt) zero_fill(e, e->kt, (size_t)8 * KT_SLOT_BLOCKS * 8 * 8);
        e->kt_layer = value; drop_graphs(e); return SONIC_OK;
    }
    if (!strcmp(key, "inject_dev_err")) {      // tests: set (1) / clear (0) the device error word a decode kernel raises when it gives up on an in-kernel wait
        HIPC(e, hipSetDevice(e->device));
        const int v = value ? 1 : 0;
        HIPC(e, hipMemcpyAsync(e->n_active + 1, &v, 4, hipMemcpyHostToDevice, e->st));
        HIPC(e, stream_sync(e));
        return SONIC_OK;
    }
    if (!strcmp(key, "no_gelu_lut")) { e->opt_no_gelu_lut = value; return SONIC_OK; }      // GELU by arithmetic instead of the LDS table (A/B)
    return fail(e, SONIC_ERR_INVALID, "unknown option %s", key);
}

// Debug read-back of an internal activation buffer as fp32 (tests / diagnostics only).
extern "C" int sonic_debug_read(sonic_engine* e, const char* name, int index, float* out, int64_t n) {
    if (!e || !name || !out) return SONIC_ERR_INVALID;
    ENTER(e);
    const sonic_dims& d = e->d;
    if (e->f32) {                                  // fp32 kind: its buffers are fp32 already
        const float* s32 = nullptr; size_t cap32 = 0;
        if (!strcmp(name, "prefill_tap")) { if (!e->taps) return fail(e, SONIC_ERR_INVALID, "no taps recorded"); s32 = (const float*)e->taps + (size_t)index * e->tok_cap * d.dec_d; cap32 = (size_t)e->tok_cap * d.dec_d; }
        else if (!strcmp(name, "pe")) { s32 = e->f->pe; cap32 = (size_t)e->Bm * e->Ta * d.dec_d; }
        else if (!strcmp(name, "dx")) { s32 = e->f->dx; cap32 = (size_t)e->tok_cap * d.dec_d; }
        else if (!strcmp(name, "enc_x")) { s32 = e->f->ln; cap32 = (size_t)e->Bm * e->T * d.enc_d; }
        else if (!strcmp(name, "h1")) { s32 = e->f->h1; cap32 = (size_t)e->Bm * (d.n_frames + 2) * d.enc_d; }
        else return fail(e, SONIC_ERR_INVALID, "unknown buffer %s", name);
        if (n < 0 || (size_t)n > cap32) return fail(e, SONIC_ERR_INVALID, "read of %lld elements exceeds buffer %s", (long long)n, name);
        HIPC(e, stream_sync(e));
        HIPC(e, d2h(e, out, s32, (size_t)n * 4));
        return SONIC_OK;
    }
    const bf16_t* src = nullptr; size_t cap = 0;
    if (!strcmp(name, "prefill_tap")) { if (!e->taps) return fail(e, SONIC_ERR_INVALID, "no taps recorded"); src = e->taps + (size_t)index * e->tok_cap * d.dec_d; cap = (size_t)e->tok_cap * d.dec_d; }
    else if (!strcmp(name, "pe")) { src = e->pe; cap = (size_t)e->Bm * e->Ta * d.dec_d; }
    else if (!strcmp(name, "dx")) { src = e->dx; cap = (size_t)e->tok_cap * d.dec_d; }
    else if (!strcmp(name, "dqkv")) { src = e->dqkv; cap = (size_t)e->tok_cap * e->qkvN; }
    else if (!strcmp(name, "dq")) { src = e->dq; cap = (size_t)e->tok_cap * e->QD; }
    else if (!strcmp(name, "datt")) { src = e->datt; cap = (size_t)e->tok_cap * e->QD; }
    else if (!strcmp(name, "dact")) { src = e->dact; cap = (size_t)e->tok_cap * d.dec_ff; }
    else if (!strcmp(name, "enc_x")) { src = e->ln; cap = (size_t)e->Bm * e->T * d.enc_d; }
    else if (!strcmp(name, "shn")) { src = e->shn; cap = (size_t)64 * d.dec_d; }          // decode-step buffers as the last step left them
    else if (!strcmp(name, "satt")) { src = e->satt; cap = (size_t)64 * e->QD; }
    else if (!strcmp(name, "sact")) { src = e->sact; cap = (size_t)64 * d.dec_ff; }
    else return fail(e, SONIC_ERR_INVALID, "unknown buffer %s", name);
    if (n < 0 || (size_t)n > cap) return fail(e, SONIC_ERR_INVALID, "read of %lld elements exceeds buffer %s", (long long)n, name);
    TmpBuf tb(e->st);
    return down_bf16(e, tb, src, out, (size_t)n);
}

// One Linear8bitLt (LLM.int8, threshold 6.0) through the engine's kernels: W [N][K] is quantised row-wise on the device, X [M][K] is
// cut into groups of `group_rows` rows (one group = one reference call: its outlier columns are found over its rows), then the int8
// MFMA GEMM with the dequantising epilogue `epi` (EPI_BIAS / _GELU / _RESID / _SWIGLU).  Inputs are fp32 holding fp16 values.
extern "C" int sonic_test_linear_int8(sonic_engine* e, const float* X, const float* W, const float* bias, const float* resid, float* out,
                                      int M, int N, int K, int group_rows, int epi) {
    if (!e || !X || !W || !out) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!e->i8) return fail(e, SONIC_ERR_INVALID, "sonic_test_linear_int8 needs an engine created with mode int8");
    if (K % 128 || N % 16 || M < 1 || group_rows < 1 || (M + group_rows - 1) / group_rows > 64)
        return fail(e, SONIC_ERR_INVALID, "bad int8 linear test shape");
    TmpBuf tb(e->st);
    const int Nout = (epi == EPI_SWIGLU) ? N / 2 : N;
    bf16_t* dX = up_bf16(e, tb, X, (size_t)M * K); bf16_t* dW = up_bf16(e, tb, W, (size_t)N * K);
    float* db = bias ? up_f32(e, tb, bias, N) : nullptr;
    bf16_t* dR = resid ? up_bf16(e, tb, resid, (size_t)M * Nout) : nullptr;
    bf16_t* dC = tb.get<bf16_t>((size_t)M * Nout);
    int8_t* cb = tb.get<int8_t>((size_t)N * K); float* scb = tb.get<float>(N);
    int8_t* qa = tb.get<int8_t>((size_t)M * K + 4096); float* sca = tb.get<float>(M);
    if (!dX || !dW || !dC || !cb || !scb || !qa || !sca) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    launch_quant_weights(dW, cb, scb, N, K, e->st);
    QW q; q.cb = cb; q.scb = scb;
    unsigned char* fl = tb.get<unsigned char>((size_t)64 * K + 64); int* occ = tb.get<int>(64); int* ocl = tb.get<int>((size_t)64 * K);
    if (!fl || !occ || !ocl) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    // the engine's own scratch is sized for its model, not for this test: swap in buffers of the test's shape for the call
    int8_t* s_qa = e->qa; float* s_sca = e->q_sca; unsigned char* s_fl = e->q_flags; int *s_occ = e->q_oc_cnt, *s_ocl = e->q_oc_list; const int s_k = e->q_kmax;
    e->qa = qa; e->q_sca = sca; e->q_flags = fl; e->q_oc_cnt = occ; e->q_oc_list = ocl; e->q_kmax = K;
    qlinear(e, epi, dX, K, nullptr, q, db, dC, Nout, M, N, K, dR, Nout, QGroup{nullptr, group_rows, (M + group_rows - 1) / group_rows});
    e->qa = s_qa; e->q_sca = s_sca; e->q_flags = s_fl; e->q_oc_cnt = s_occ; e->q_oc_list = s_ocl; e->q_kmax = s_k;
    return down_bf16(e, tb, dC, out, (size_t)M * Nout);
}

// greedy_kernel on caller-provided lm_head partial slabs [ksplit][mpad][V] (fp32): returns the token each row picks (first maximum of
// the bf16-rounded slab sum, HF:generation/utils.py:2925 / torch.argmax semantics) and, optionally, the bf16 logits it compared.
extern "C" int sonic_test_greedy(sonic_engine* e, const float* slabs, int ksplit, int mpad, int V, int B, int32_t* tok_out, float* logits_out) {
    if (!e || !slabs || !tok_out) return SONIC_ERR_INVALID;
    ENTER(e);
    if (ksplit < 1 || ksplit > 8 || B < 1 || B > 64 || mpad < B || V < 4 || V % 4) return fail(e, SONIC_ERR_INVALID, "bad greedy test shape");
    TmpBuf tb(e->st);
    const size_t n = (size_t)ksplit * mpad * V;
    float* dl = up_f32(e, tb, slabs, n);
    bf16_t* table = tb.get<bf16_t>((size_t)V * 8); bf16_t* x = tb.get<bf16_t>((size_t)64 * 8);
    int* st = tb.get<int>(64 * 8 + 4); int* ids = tb.get<int>(64);
    float* dump = logits_out ? tb.get<float>((size_t)B * V) : nullptr;
    if (!dl || !table || !x || !st || !ids || (logits_out && !dump)) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    std::vector<int> h(64 * 8 + 4, 0);
    for (int b = 0; b < 64; ++b) { h[64 * 2 + b] = 1; h[64 * 4 + b] = 4; }       // kv_len = 1, max_new = 4
    h[64 * 8] = B;
    HIPC(e, h2d(e, st, h.data(), h.size() * 4));
    GreedyArgs g{};
    g.logits = dl; g.ksplit = ksplit; g.mpad = mpad; g.V = V; g.B = B; g.table = table; g.x = x; g.d = 8;
    g.out_ids = ids; g.out_ld = 1; g.n_new = st; g.finished = st + 64; g.kv_len = st + 128; g.tok_pos = st + 192; g.max_new = st + 256;
    g.n_active = st + 512; g.n_eos = 0; g.pad_id = 0; g.logits_dump = dump; g.dump_stride_step = (long)B * V; g.step_counter = dump ? st + 320 : nullptr;
    launch_greedy(g, e->st);
    HIPC(e, stream_sync(e));
    HIPC(e, hipGetLastError());
    std::vector<int> out(64);
    HIPC(e, d2h(e, out.data(), ids, 64 * 4));
    for (int b = 0; b < B; ++b) tok_out[b] = out[b];
    if (logits_out) HIPC(e, d2h(e, logits_out, dump, (size_t)B * V * 4));
    return SONIC_OK;
}

extern "C" int sonic_test_skinny_gu(sonic_engine* e, const float* X, const float* Wgu_interleaved, float* act, int M, int N, int K) {
    if (!e) return SONIC_ERR_INVALID;
    ENTER(e);
    if (!skinny_gu_eligible(M, N, K)) return fail(e, SONIC_ERR_INVALID, "shape not handled by the fused gate/up kernel");
    TmpBuf tb(e->st);
    bf16_t* dX = up_bf16(e, tb, X, (size_t)M * K); bf16_t* dW = up_bf16(e, tb, Wgu_interleaved, (size_t)N * K);
    bf16_t* dWt = tb.get<bf16_t>((size_t)N * K); bf16_t* dA = tb.get<bf16_t>((size_t)M * (N / 2));
    if (!dX || !dW || !dWt || !dA) return fail(e, SONIC_ERR_OOM, "HIP out of memory in test hook");
    launch_tile_weights_gu8(dW, dWt, N, K, e->st);
    SkinnyArgs a{}; a.X = dX; a.ldx = K; a.W = dWt; a.M = M; a.N = N; a.K = K; a.ksplit = 1; a.dt = e->dt;
    launch_skinny_gu(a, dA, e->st);
    return down_bf16(e, tb, dA, act, (size_t)M * (N / 2));
}
