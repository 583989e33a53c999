/*
 * sonic_hip.h -- C ABI of libsonic_hip.so, the MI355X-native engine behind SonicScribe's
 * ASRModel.transcribe() (reference: backend/asr.py:335-488).
 *
 * The reference has no FFI of its own (pure Python); this is the boundary its
 * `ASRModel` methods bind through ctypes (sonicscribe_amd/engine.py).  Plain
 * pointers and sizes only; every function returns a sonic_status; outputs are caller-allocated;
 * no callbacks; no global state besides the handle.  Calls on one engine are serialised
 * internally (the reference is entered from up to 3 executor threads plus the event-loop
 * thread, backend/main.py:429-430,616-624, backend/transcription_manager.py:58).
 *
 * Which reference interface each entry point replaces:
 *   sonic_create / sonic_load_* / sonic_finalize_weights
 *                              ASRModel.__init__ + _load_model_standard   asr.py:25-87,120-146
 *                              (models_manager.asr_model_init             models_manager.py:16-32)
 *   sonic_transcribe_batch     the device part of ASRModel.transcribe     asr.py:393-422
 *                              (processor features + model.generate(do_sample=False))
 *   sonic_stage_pcm / sonic_run_staged / sonic_fetch_tokens
 *                              the same, split so PCM can be HBM-resident before a timed region
 *   sonic_logmel               WhisperFeatureExtractor.__call__ as invoked from asr.py:393
 *                              (HF:feature_extraction_whisper.py:193-346)
 *   sonic_encode               GlmAsrModel.get_audio_features             HF:modeling_glmasr.py:380-408
 *   sonic_prefill / sonic_decode_step
 *                              the two halves of model.generate: the prompt forward (logits_to_keep = 1,
 *                              HF:generation/utils.py:2612-2616) and iterations of the greedy loop (:2876-2943)
 *   sonic_device_info          torch.version.cuda / torch.cuda.get_device_name() / get_device_properties(0).total_memory
 *                              in ASRModel.get_model_info                   asr.py:501-506
 *   sonic_memory_info          torch.cuda.memory_allocated() / memory_reserved() in the debug dict  asr.py:453-457
 *   sonic_destroy              `del asr_model.model` + torch.cuda.empty_cache(): every device byte goes back   backend/main.py:84-90
 *   sonic_last_error           the exception text re-raised at            asr.py:469-481
 */
#ifndef SONIC_HIP_H
#define SONIC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: these declarations are its whole dynamic symbol table (tests/test_host_logic.py checks
 * `nm -D` against this header).  SONIC_ABI_VERSION moves whenever a signature or a struct layout below changes. */
#define SONIC_API __attribute__((visibility("default")))
#define SONIC_ABI_VERSION 7
SONIC_API int sonic_abi_version(void);

typedef struct sonic_engine sonic_engine;

typedef enum {
    SONIC_OK = 0,
    SONIC_ERR_INVALID = 1,   /* bad argument / shape / state            -> ValueError / RuntimeError */
    SONIC_ERR_HIP = 2,       /* HIP runtime failure                     -> RuntimeError */
    SONIC_ERR_OOM = 3,       /* message contains "out of memory" (asr.py:471 hint branch) */
    SONIC_ERR_MISMATCH = 4,  /* audio placeholders != audio feature rows (HF:modeling_glmasr.py:426-429) */
    SONIC_ERR_UNSUPPORTED = 5
} sonic_status;

/* SONIC_MODE_INT8 = asr.py mode="int8" (:148-210): fp16 activations, every nn.Linear except lm_head / embed_tokens replaced by
 * LLM.int8() (row-wise int8 weights, per-token int8 activations, outlier columns |x| >= 6.0 in fp16).  bitsandbytes is absent offline,
 * so this mode is checked against the restatement in oracle/sonic_oracle.c only (parity unpinned). */
enum { SONIC_MODE_NATIVE = 0 /* bf16, asr.py mode="native" */, SONIC_MODE_INT8 = 1 /* asr.py mode="int8" */,
       /* test only: fp16 weights and activations through the SAME kernel templates (their KF16 / f16_t instantiations - since round 5 also
        * the fused decode kernels skinny_o / skinny_gu the bf16 headline runs), no quantisation.  Not an asr.py mode: it exists so that layouts,
        * epilogues, RoPE, masking, KV append and the greedy controller can be checked at fp16's 8x finer rounding against the reference's fp32
        * arithmetic (tests/test_gpu_fp16_mode.py) */
       SONIC_MODE_F16 = 2,
       /* test only: the fp32 KIND of every stage (csrc/f32kind.hip) - fp32 weights, activations and accumulation behind the same request plan, PCM
        * staging, log-mel kernel, prompt assembly, KV bookkeeping and greedy controller (greedy_kernel<float>).  It exists to show north_star's
        * "within 1e-3 on logits" literally on the GPU (tests/test_gpu_fp32_mode.py against tests/golden/\*_fp32.npz); the product kinds keep the
        * reference's bf16 / fp16 op-boundary roundings and cannot.  No slots, no continuous decoding, no graphs; speed is irrelevant. */
       SONIC_MODE_F32 = 3 };
enum { SONIC_DTYPE_F32 = 0, SONIC_DTYPE_BF16 = 1 };

/* Model dimensions (defaults: HF:configuration_glmasr.py:44-54,86-103). */
typedef struct {
    int32_t n_mels, n_frames, enc_T;
    int32_t enc_d, enc_ff, enc_layers, enc_heads, enc_rotary_dim;
    float enc_theta, enc_ln_eps;
    int32_t merge;
    int32_t dec_d, dec_ff, dec_layers, dec_heads, dec_kv_heads, dec_head_dim;
    float dec_theta, dec_rms_eps;
    int32_t vocab, audio_token_id, n_eos;
    int32_t eos[8];
} sonic_dims;

/* Per-stage device time of the last sonic_run_staged / sonic_transcribe_batch (HIP events on the engine stream). */
typedef struct {
    float mel_ms, encoder_ms, prefill_ms, decode_ms, total_ms;
    float gemm_ms;          /* summed duration of the encoder's dominant GEMM kernel launches (fc1, bias+GELU epilogue) */
    int32_t gemm_launches;  /* ... and how many launches that was */
    double gemm_flops;      /* algorithmic FLOPs of those launches */
    int32_t decode_steps;
    float enc_gemm_ms;      /* summed duration of ALL encoder-layer GEMM launches (QKV, o, fc1, fc2) of the last run */
    double enc_gemm_flops;  /* ... and their algorithmic FLOPs (SURVEY.md 8d "encoder GEMM MFMA utilisation") */
    /* host side of the same run (wall clock of the calling / worker thread): enqueueing everything up to the first token, time inside the
     * decode loop's hipGraphLaunch calls, time blocked on the pipelined early-stop checks, and how many chunk launches that was */
    float host_prefill_enqueue_ms, host_decode_launch_ms, host_decode_wait_ms;
    int32_t host_decode_launches;
    int32_t decode_lookahead;   /* chunks the decode loop kept queued beyond the early-stop check it was waiting for (adapts: 1 on a host that keeps up) */
    int32_t decode_launches_per_layer;   /* kernel launches per decoder layer of the token step the last run used (5: <= 2 rows, round 6; 6: fused bf16 / fp16 chain; 7: 33 - 64
                                          * rows inside continuous loops; 8: int8 and the unfused chain) */
} sonic_timings;

/* ---- lifetime ---- */
SONIC_API int sonic_device_count(void);
/* max_batch: windows per call (<= 64); max_ctx: decoder context capacity per sequence (multiple of 64). */
SONIC_API int sonic_create(const sonic_dims* dims, int device_id, int mode, int max_batch, int max_ctx, sonic_engine** out);
SONIC_API void sonic_destroy(sonic_engine* e);
SONIC_API const char* sonic_last_error(sonic_engine* e); /* e may be NULL: error of the last failed sonic_create on this thread */
/* name (NUL-terminated, truncated to name_cap), total / currently free device memory, hipRuntimeGetVersion(); any output may be NULL */
SONIC_API int sonic_device_info(int device_id, char* name, int name_cap, int64_t* total_bytes, int64_t* free_bytes, int32_t* hip_runtime_version);
/* How many hardware queues the HIP runtime of THIS process gives its streams on the device (measured once per device: eight probe streams with a
 * 300 us spin kernel each; streams that share a queue run in order), what GPU_MAX_HW_QUEUES says now (0 = unset) and how many an engine with slots
 * wants (8).  The runtime reads the variable at its first call only: in the reference's process torch initialises it (backend/asr.py:53
 * `torch.cuda.is_available()`), so the host sets GPU_MAX_HW_QUEUES=8 in its environment - the library never writes the
 * environment, and the first sonic_create warns on stderr when the measured count is below the wanted one (SONIC_QUIET=1 silences it). */
SONIC_API int sonic_runtime_info(int device_id, int32_t* hw_queues, int32_t* hw_queues_env, int32_t* hw_queues_wanted);
/* allocated: bytes of this handle's live device allocations (a slot: its own buffers; the weights are its owner's); reserved = allocated
 * (no caching allocator under the engine: sonic_destroy returns everything to the driver) */
SONIC_API int sonic_memory_info(sonic_engine* e, int64_t* allocated_bytes, int64_t* reserved_bytes);

/* Another batch in flight on the SAME weight copy.  The reference keeps up to three decodes in flight on its one model object in file
 * mode (backend/main.py:429-445, asyncio.Semaphore(3) + run_in_executor at :616-624) and serialises them on the device; here a slot is a
 * full engine handle of its own -- stream, PCM staging, activation buffers, KV cache, decode graphs, lock, options -- whose weight and
 * constant pointers are the owner's (sonic_weight_bytes(owner) does not move, sonic_weight_bytes(slot) == 0).  Batches on different slots
 * run concurrently on the GPU: one batch's MFMA-bound encoder / prefill fills the bubbles of another's latency-bound decode loop
 * (DESIGN.md 4).  Every entry point below takes a slot handle; rings created through any handle can be staged by every slot of the same
 * owner.  sonic_destroy(slot) releases a slot early; sonic_destroy(owner) releases its slots first (their handles are dead after that).
 * sonic_slot_count: the owner plus its live slots. */
SONIC_API int sonic_slot_create(sonic_engine* owner_or_slot, sonic_engine** slot_out);
SONIC_API int sonic_slot_count(sonic_engine* e);
/* row / context capacity, mode, device and the identity of the weight copy (equal for an engine and all of its slots) of a handle; any out pointer may be NULL */
SONIC_API int sonic_engine_info(sonic_engine* e, int32_t* max_batch, int32_t* max_ctx, int32_t* mode, int32_t* device_id, const void** weights_id);

/* ---- the bulk pipeline as native threads (round 5) ----
 * What sonicscribe_amd/pipeline.py's host loop does (round 4: Python threads over the calls above, polling) inside the library: one thread per
 * prefill handle (stage PCM, queue log-mel + encoder + prompt forward + first token, hand the batch over), one per decoding handle (splice
 * handed-over batches into free row blocks of its continuously decoding handle, queue decode chunks, fetch the rows of a block the moment all of
 * them are finished).  Every wait is a condition variable or a blocking HIP event: no polling, no interpreter in the loop - the headline no longer
 * depends on how quickly a busy host schedules Python threads.  The reference's counterpart: three executor threads around one model object
 * (backend/main.py:429-445, 616-624).
 *   sonic_pipeline_create(decoders, n_dec, prefills, n_pre, block, rows_per_decoder, &p)   handles of ONE weight copy (an engine and its slots);
 *                       the decoders are put into continuous mode (sonic_service_begin); a decoder holds rows_per_decoder / block batches at a time
 *   sonic_pipeline_submit(p, pcm, offsets, W, req_win, R, prompt_ids, prompt_off, max_new, out_ids, out_ld, out_len, &ticket)
 *                       one batch of R <= block requests (arguments as sonic_transcribe_batch).  pcm == NULL: the batch is what the prefill
 *                       handles already have staged (sonic_stage_pcm on each of them; the benchmark's HBM-resident input).  The prompt arrays
 *                       are copied; pcm, out_ids and out_len must stay valid until the ticket is waited for.  Returns at once.
 *   sonic_pipeline_wait(p, ticket)   blocks until that batch's rows are in out_ids / out_len and returns its status; ticket 0: every batch
 *                       submitted so far, first failure or SONIC_OK.  A bad request fails alone; a failed decoding handle fails everything in flight.
 *   sonic_pipeline_destroy(p)       completes what was submitted, joins the threads, sonic_service_end on the decoders (the handles stay yours) */
typedef struct sonic_pipeline sonic_pipeline;
SONIC_API int sonic_pipeline_create(sonic_engine* const* decoders, int n_dec, sonic_engine* const* prefills, int n_pre, int block, int rows_per_decoder,
                                    sonic_pipeline** out);
SONIC_API int sonic_pipeline_submit(sonic_pipeline* p, const int16_t* pcm, const int64_t* offsets, int W, const int32_t* req_win, int R,
                                    const int32_t* prompt_ids, const int64_t* prompt_off, const int32_t* max_new,
                                    int32_t* out_ids, int out_ld, int32_t* out_len, int64_t* ticket_out);
SONIC_API int sonic_pipeline_wait(sonic_pipeline* p, int64_t ticket);
SONIC_API int sonic_pipeline_stats(sonic_pipeline* p, int64_t* batches_done, int64_t* chunks_queued, int32_t* batches_in_flight_max);
SONIC_API const char* sonic_pipeline_last_error(sonic_pipeline* p);
SONIC_API int sonic_pipeline_destroy(sonic_pipeline* p);

/* ---- weights (names: GlmAsrForConditionalGeneration.state_dict() keys, see sonicscribe_amd/spec.py) ---- */
SONIC_API int sonic_load_tensor(sonic_engine* e, const char* name, const void* data, int dtype, const int64_t* shape, int ndim);
SONIC_API int sonic_load_synthetic(sonic_engine* e, uint64_t seed);          /* portable generator, sonicscribe_amd/synth.py */
SONIC_API int sonic_finalize_weights(sonic_engine* e);                        /* packs fused QKV / gate-up, conv im2col order */
SONIC_API int64_t sonic_weight_bytes(sonic_engine* e);

/* ---- stage entry points (parity tests) ---- */
/* pcm: B segments concatenated; offsets[B+1] in samples (segment i = pcm[offsets[i] .. offsets[i+1])), each <= 30 s.
 * feats_out: [B][n_mels][n_frames] fp32 (HF layout), mask_out: [B][n_frames] int32; either may be NULL. */
SONIC_API int sonic_logmel(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int B, float* feats_out, int32_t* mask_out);
/* feats: [B][n_mels][n_frames] fp32 (cast to bf16 as asr.py:280-301 does); n_valid_frames[B].
 * embeds_out: [B][enc_T/merge][dec_d] fp32 (first n_audio_out[b] rows valid).
 * taps (optional, may be NULL): enc_layers_out [B][enc_layers][enc_T][enc_d], enc_out [B][enc_T][enc_d]. */
SONIC_API int sonic_encode(sonic_engine* e, const float* feats, const int32_t* n_valid_frames, int B,
                 float* embeds_out, int32_t* n_audio_out, float* enc_layers_out, float* enc_out);

/* ---- the hot call ---- */
/* W windows of PCM (as sonic_logmel); R requests, request r owns windows req_win[r] .. req_win[r+1]-1 (R == W and
 * req_win == NULL for the single-window case).  prompt_ids concatenated, prompt_off[R+1]; max_new[R].
 * out_ids: [R][out_ld] int32, out_len[R]; step_logits (optional): [max(max_new)][R][vocab] fp32 = the bf16 logits
 * each step's argmax saw (row r of step s valid while s < out_len[r]). */
SONIC_API int sonic_transcribe_batch(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int W,
                           const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                           const int32_t* max_new, int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits);

/* split form: stage (H2D) -> run (device only, timed) -> fetch (D2H) */
SONIC_API int sonic_stage_pcm(sonic_engine* e, const int16_t* pcm, const int64_t* offsets, int W);
SONIC_API int sonic_run_staged(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                     const int32_t* max_new, int want_step_logits);
SONIC_API int sonic_fetch_tokens(sonic_engine* e, int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits);
/* sonic_run_staged without blocking the caller: the arguments are copied, a worker thread owned by the handle runs the batch, the call
 * returns at once (the reference's counterpart is loop.run_in_executor(None, asr_model.transcribe, ...), backend/main.py:616-624).  One
 * outstanding run per handle; until sonic_wait has returned the handle takes no other call (ring appends excepted).
 * sonic_wait(e, 1, NULL) blocks until the run is complete and returns ITS status; sonic_wait(e, 0, &busy) polls. */
SONIC_API int sonic_run_staged_async(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                                     const int32_t* max_new, int want_step_logits);
SONIC_API int sonic_wait(sonic_engine* e, int block, int32_t* busy_out);
/* stage entry points: sonic_run_staged = sonic_prefill + sonic_decode_step(max(max_new) - 1).  sonic_prefill runs log-mel, encoder,
 * projector, decoder prefill and emits the first token of every request; sonic_decode_step runs up to n_steps further greedy steps
 * (*steps_done_out of them: fewer once the largest budget is reached or every row stopped) and reports the rows still running.
 * sonic_fetch_tokens may be called after either. */
SONIC_API int sonic_prefill(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off,
                  const int32_t* max_new, int want_step_logits);
SONIC_API int sonic_decode_step(sonic_engine* e, int n_steps, int32_t* n_active_out, int32_t* steps_done_out);
/* sonic_prefill without the closing wait: the work is queued on the handle's stream when the call returns (the splice of its rows into a
 * continuously decoding handle orders itself behind it on the device; any other reader calls sonic_synchronize first) */
SONIC_API int sonic_prefill_enqueue(sonic_engine* e, const int32_t* req_win, int R, const int32_t* prompt_ids, const int64_t* prompt_off, const int32_t* max_new);
/* Continuous decoding: one handle decodes forever over its max_batch rows; requests join and leave ROW BY ROW instead of batch by batch.  The
 * reference awaits one transcribe() at a time per connection and blocks its event loop inside it (backend/connection_manager.py:127-245,
 * backend/transcription_manager.py:58); a batch engine makes a request wait for the running batch and pads every batch to its slowest row.
 *   sonic_service_begin(d)      d's rows become a pool (all free), its chunk graph is captured; d takes no batch calls until sonic_service_end
 *   sonic_prefill(p, ...)       on ANOTHER handle of the same weights (a slot): log-mel, encoder, prompt forward, first token of R requests
 *   sonic_splice_rows(d, p, n, src_rows, dst_rows, &seq)   rows src_rows[] of p (KV cache, control words, next-step input) -> free rows
 *                               dst_rows[] of d, queued on d's stream between two chunks; p may start its next prefill at once (it waits for the
 *                               copies on the device).  seq = chunks d had queued before: checks with a larger number describe the new occupants
 *   sonic_service_step(d, k, rows, finished[64], n_new[64], &seq, &n_active)   queue k more chunks (k * decode_chunk token steps for rows
 *                               0 .. rows-1 rounded up to 16; rows = 0: all - the caller names the highest occupied row + 1, so a lightly loaded
 *                               pool steps faster) and return the newest completed check: finished[r] = 1 once row r hit EOS / its budget (or is
 *                               free), n_new[r] its tokens
 *   sonic_fetch_row(d, row, n, ids)   the n tokens of a finished row; the row is free again
 *   sonic_fetch_rows(d, n, rows, counts, ids, ld)   the same for n finished rows in one call (row i's tokens at ids + i * ld)
 * A request's tokens are the same bits as in a solo run (rows are independent in every decode kernel; tests/test_gpu_continuous.py). */
SONIC_API int sonic_service_begin(sonic_engine* d);
SONIC_API int sonic_service_end(sonic_engine* d);
SONIC_API int sonic_splice_rows(sonic_engine* d, sonic_engine* p, int n, const int32_t* src_rows, const int32_t* dst_rows, int64_t* seq_out);
SONIC_API int sonic_service_step(sonic_engine* d, int n_chunks, int rows, int32_t* finished_out, int32_t* n_new_out, int64_t* seq_out, int32_t* n_active_out);
SONIC_API int sonic_fetch_row(sonic_engine* d, int row, int n, int32_t* out_ids);
SONIC_API int sonic_fetch_rows(sonic_engine* d, int n, const int32_t* rows, const int32_t* counts, int32_t* out_ids, int out_ld);

/* Request-level scheduling for live traffic as native threads (csrc/dispatch.cpp): what the reference does with one `await
 * transcribe()` per partial / final of every WebSocket session (backend/connection_manager.py:127-245, backend/transcription_manager.py:19-65) and
 * three executor threads in file mode (backend/main.py:429-445, 616-624), all serialised on one model object.  decoders: handles that decode
 * continuously over their max_batch rows (sonic_service_begin is called on them); prefills: handles that run log-mel + encoder + prompt forward +
 * first token of whatever is queued - as many requests as the emptiest decoder has free rows - and hand the rows over; all handles share one
 * weight copy.  A request is W windows (host PCM, or slices of device rings: see sonic_stage_mixed), a prompt and a budget; it joins a running loop
 * between two chunks and leaves the moment it hits EOS / its budget.  Tokens equal the solo run's bit for bit (decode rows are independent). */
typedef struct sonic_dispatch sonic_dispatch;
struct sonic_ring;
SONIC_API int sonic_dispatch_create(sonic_engine* const* decoders, int n_dec, sonic_engine* const* prefills, int n_pre, int adaptive_tiles, sonic_dispatch** out);
SONIC_API int sonic_dispatch_submit(sonic_dispatch* d, const int16_t* host_pcm, const int64_t* host_off, struct sonic_ring* const* rings, const int64_t* ring_start,
                                    const int32_t* ring_n, int W, const int32_t* prompt_ids, int prompt_len, int max_new, int64_t* ticket_out);
SONIC_API int sonic_dispatch_cancel(sonic_dispatch* d, int64_t ticket);                 /* queued requests only */
/* next completed request in completion order; blocks up to timeout_ms (< 0: until one completes or the dispatcher is closed and drained); *ticket_out = 0: none */
SONIC_API int sonic_dispatch_next(sonic_dispatch* d, int timeout_ms, int64_t* ticket_out, int32_t* status_out, int32_t* out_ids, int out_cap, int32_t* n_out,
                                  char* err, int err_cap);
SONIC_API int sonic_dispatch_stats(sonic_dispatch* d, int64_t* prefill_batches, int64_t* decode_chunks, int32_t* load_windows, int32_t* free_rows);
SONIC_API int sonic_dispatch_close(sonic_dispatch* d);     /* queued requests fail, running ones complete (still collectable); handles leave continuous mode */
SONIC_API int sonic_dispatch_destroy(sonic_dispatch* d);

/* Device-resident ingest.  A ring holds the raw wire PCM of one streaming session in HBM: what the reference keeps as
 * 2048-byte chunks in a host dict (backend/audio_manager.py:21-33, fed from backend/main.py:813-842) and concatenates on the host for
 * every partial / final decode (audio_manager.py:99-123).  A decode names sample ranges of rings instead of handing over host buffers;
 * the reference's conversions between the wire and the feature extractor -- int16 -> float32 / 32768
 * (backend/transcription_manager.py:45-54), peak normalisation over the request and the PCM_16 round trip (backend/asr.py:247-276) --
 * run on the device, bit-identical with the host path.  Appends use the ring's own stream and lock: they do not wait for a batch that
 * is decoding.  A ring belongs to the engine it was created on; sonic_destroy frees the rings that are still alive (their handles
 * are dead after that). */
typedef struct sonic_ring sonic_ring;
SONIC_API int sonic_ring_create(sonic_engine* e, int64_t capacity_samples, sonic_ring** out);
SONIC_API void sonic_ring_destroy(sonic_ring* r);
/* append n samples (any chunk size <= capacity); *first_index = absolute sample index of pcm[0].  Returns at once: the samples are
 * copied to a pinned mirror and their H2D copy is queued; decodes that name them order behind it */
SONIC_API int sonic_ring_append(sonic_ring* r, const int16_t* pcm, int64_t n, int64_t* first_index);
SONIC_API int64_t sonic_ring_head(sonic_ring* r);     /* samples appended so far */
/* sonic_transcribe_batch with every window either host samples (rings == NULL or rings[w] == NULL: int16 PCM already peak-normalised,
 * host_off[W+1]; ring windows have empty host ranges) or samples [ring_start[w], ring_start[w] + ring_n[w]) of rings[w], which must
 * still be inside the ring's last `capacity` samples.  The windows of one request (req_win) share one peak.  Without req_win R == W. */
SONIC_API int sonic_transcribe_mixed(sonic_engine* e, const int16_t* host_pcm, const int64_t* host_off, sonic_ring* const* rings,
                           const int64_t* ring_start, const int32_t* ring_n, int W, const int32_t* req_win, int R,
                           const int32_t* prompt_ids, const int64_t* prompt_off, const int32_t* max_new,
                           int32_t* out_ids, int out_ld, int32_t* out_len, float* step_logits);
/* the staging half alone (then sonic_run_staged / sonic_fetch_tokens) */
SONIC_API int sonic_stage_mixed(sonic_engine* e, const int16_t* host_pcm, const int64_t* host_off, sonic_ring* const* rings,
                      const int64_t* ring_start, const int32_t* ring_n, int W, const int32_t* req_win, int R);

/* Teacher forcing (parity tests; mirrors the oracle's force_ids): while set, token n of request r is ids[r * ld + n] instead of
 * the argmax -- logits are still computed and returned, EOS / budget rules apply to the forced token (HF:generation/utils.py:2925-2936
 * with next_tokens replaced).  ids == NULL clears.  Forced runs use the eager decode loop. */
SONIC_API int sonic_set_forced_ids(sonic_engine* e, const int32_t* ids, int R, int ld);
SONIC_API int sonic_get_timings(sonic_engine* e, sonic_timings* out);
SONIC_API int sonic_synchronize(sonic_engine* e);

/* ---- single-kernel test hooks (host fp32 in/out, converted to bf16 on device; used by tests/ only) ---- */
SONIC_API int sonic_test_gemm(sonic_engine* e, const float* A, const float* W, const float* bias, const float* resid, float* C,
                    int M, int N, int K, int epi);
SONIC_API int sonic_test_skinny(sonic_engine* e, const float* X, const float* W, float* C, int M, int N, int K);
SONIC_API int sonic_test_skinny_gu(sonic_engine* e, const float* X, const float* Wgu_interleaved, float* act, int M, int N, int K);
/* the argmax + greedy controller on caller-provided lm_head partial slabs [ksplit][mpad][V] fp32: token picked per row
 * (first maximum of the bf16-rounded slab sum) and optionally the bf16 logits [B][V] it compared */
/* one Linear8bitLt call (backend/asr.py:182-198: bnb.nn.Linear8bitLt(has_fp16_weights=False, threshold=6.0)) through the engine's int8
 * kernels: W [N][K] quantised row-wise on the device, X [M][K] in groups of group_rows rows (one group = one reference call), int8 MFMA
 * GEMM + dequantising epilogue epi.  fp32 buffers holding fp16 values.  Needs an engine created with SONIC_MODE_INT8. */
SONIC_API int sonic_test_linear_int8(sonic_engine* e, const float* X, const float* W, const float* bias, const float* resid, float* out,
                           int M, int N, int K, int group_rows, int epi);
SONIC_API int sonic_test_greedy(sonic_engine* e, const float* slabs, int ksplit, int mpad, int V, int B, int32_t* tok_out, float* logits_out);
SONIC_API int sonic_test_attention(sonic_engine* e, const float* q, const float* k, const float* v, float* out,
                         int B, int Tq, int Tk, int Hq, int Hkv, int hd, int causal);
SONIC_API int sonic_test_decode_attention(sonic_engine* e, const float* q, const float* k, const float* v, float* out,
                                int B, int Tk, int Hq, int Hkv);
SONIC_API int sonic_test_layernorm(sonic_engine* e, const float* x, const float* w, const float* b, float* y, int rows, int d, float eps, int rms);
/* times `iters` launches of the encoder's dominant GEMM shape on the engine stream with HIP events */
SONIC_API int sonic_bench_gemm(sonic_engine* e, int M, int N, int K, int epi, int iters, float* ms_per_launch);
/* times the decode-step skinny GEMM (variant: 0 LDS-DMA nt, 1 LDS-DMA default policy, 2 registers nt, 3 registers plain, 9 pure-read floor) */
SONIC_API int sonic_bench_skinny(sonic_engine* e, int M, int N, int K, int variant, int iters, float* us_per_launch);
/* debug read-back of an internal bf16 activation buffer as fp32 ("prefill_tap" with index = 0 (embeddings) .. dec_layers,
 * "pe", "dx", "dqkv", "dq", "datt", "dact", "enc_x"); tests / diagnostics only */
SONIC_API int sonic_debug_read(sonic_engine* e, const char* name, int index, float* out, int64_t n);
/* diagnostics: in-kernel timestamps (100 MHz device wall clock) of the decode kernels of one decoder layer, recorded while the option
 * "ktrace" = layer index is set: out[slot][block < 512][8 points], slots 0 qkv, 1 attention, 2 o_proj, 3 gate/up, 4 down */
SONIC_API int sonic_debug_ktrace(sonic_engine* e, int64_t* out, int64_t n);
/* per-engine experiment knobs: "skinny_variant", "gemm_force128", "gemm256_stagger", "prefill_taps", "no_fused_gu",
 * "no_graph" (eager decode loop), "decode_chunk" (token steps per hipGraph launch = granularity of the early-stop check and of row splices, default 2),
 * "decode_lookahead" (start value of the adaptive queue depth of the decode loop, in chunks), "gemm_timing" (HIP events around every encoder-layer GEMM launch -> sonic_timings.enc_gemm_*),
 * "no_fused_rope" (encoder RoPE as its own pass), "no_gelu_lut" (fc1 GELU by arithmetic instead of the LDS table); the full list with what each one measured is in
 * DESIGN.md 1.  Round 6: "no_pre_norm" (<= 2 rows: standalone add + RMSNorm launches instead of the five-launch chain; same bits), "decode_gemv" / "decode_prefetch" /
 * "decode_attn_occ2" (experiments that lost: profiles/round6_*), "f32_synth_bf16" (SONIC_MODE_F32: sonic_load_synthetic writes the bf16-rounded values),
 * "inject_dev_err" (tests: sets / clears the device error word) */
SONIC_API int sonic_set_option(sonic_engine* e, const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* SONIC_HIP_H */
