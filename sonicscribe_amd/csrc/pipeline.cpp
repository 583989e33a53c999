// sonic_pipeline_*: the bulk pipeline's hand-overs as native threads.
//
// What it replaces: sonicscribe_amd/pipeline.py's host loop (Python threads over ctypes: a prefiller per slot, a decoder thread per decoding
// handle, polling with 2 / 20 ms sleeps) - the reason the headline leg lost 10-13 % beside a busy host while the natively queued legs lost
// < 2 % (profiles/round4_busy_host_ab.txt).  In the reference the same role is played by three executor threads around one model object
// (backend/main.py:429-445, 616-624).  Here:
//   * one thread per prefill handle: takes the next submitted batch, stages its PCM (or uses what the handle has staged), QUEUES log-mel +
//     encoder + prompt forward + first token (sonic_prefill_enqueue) and hands the batch over; it blocks on a condition variable until a decoder
//     has queued the splice of its rows (the handle's buffers are the splice's source until then);
//   * one thread per decoding handle: splices handed-over batches into free row blocks of its continuously decoding handle, queues decode
//     chunks (sonic_service_step: blocking, sleeps on an event), fetches the rows of a block the moment the pipelined check shows every row of
//     it finished, completes the batch's ticket.  An EMPTY loop leaves the next batch to a loop that is running part-filled (two batches in one
//     64-row loop stream the weights once).
// No thread polls: every wait is a condition variable or a blocking HIP event inside the engine.  Only the C ABI of include/sonic_hip.h is used.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sonic_hip.h"

namespace {

struct Batch {
    int64_t ticket = 0;
    const int16_t* pcm = nullptr; std::vector<int64_t> offsets; int W = 0;      // pcm == nullptr: what the prefill handle has staged
    std::vector<int32_t> req_win; bool has_req_win = false;
    int R = 0;
    std::vector<int32_t> prompt_ids; std::vector<int64_t> prompt_off; std::vector<int32_t> max_new;
    int32_t* out_ids = nullptr; int out_ld = 0; int32_t* out_len = nullptr;
    int status = -1;                                    // -1 running, else a sonic_status
    std::string err;
};
struct Ready { std::shared_ptr<Batch> b; sonic_engine* src = nullptr; bool taken = false; };

}   // namespace

struct sonic_pipeline {
    std::vector<sonic_engine*> dec, pre;
    int block = 32, blocks_per_dec = 1;
    bool pair = true;
    std::mutex mu;
    std::condition_variable cv;                          // one for everything: submissions, hand-overs, completions, state changes
    std::deque<std::shared_ptr<Batch>> queue;            // submitted, not yet taken by a prefill thread
    std::deque<std::shared_ptr<Ready>> ready;            // prefill queued, waiting for a decoder
    std::vector<std::shared_ptr<Batch>> all;             // every batch not yet collected by sonic_pipeline_wait (by ticket)
    std::vector<char> half;                              // per decoder: it has both running and free blocks
    int64_t next_ticket = 1, done_batches = 0, chunks = 0;
    int64_t chunks_full = 0, chunks_part = 0; double prefill_wait_ms = 0, decoder_idle_ms = 0;   // diagnostics (SONIC_PIPE_STATS=1: printed by sonic_pipeline_destroy)
    bool stop = false;
    int failed = 0; std::string fail_msg;                // a decoding handle failed: nothing can complete any more
    std::string last_err;
    std::vector<std::thread> threads;
    bool began = false;
};

namespace {

void finish(sonic_pipeline* p, const std::shared_ptr<Batch>& b, int status, const std::string& err) {   // p->mu held
    if (b->status != -1) return;
    b->status = status; b->err = err;
    ++p->done_batches;
    p->cv.notify_all();
}

void prefill_thread(sonic_pipeline* p, sonic_engine* h) {
    for (;;) {
        std::shared_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || p->failed || !p->queue.empty(); });
            if (p->failed) {
                // a decoding handle failed: submits are refused from now on (sonic_pipeline_submit returns p->failed), so what is queued is all there
                // will ever be - fail it and sleep until destroy (the predicate above stays true, a `continue` here would spin)
                while (!p->queue.empty()) { finish(p, p->queue.front(), p->failed, p->fail_msg); p->queue.pop_front(); }
                p->cv.wait(lk, [&] { return p->stop; });
                return;
            }
            if (p->queue.empty()) { if (p->stop) return; continue; }
            b = p->queue.front(); p->queue.pop_front();
        }
        int rc = SONIC_OK;
        if (b->pcm) rc = sonic_stage_pcm(h, b->pcm, b->offsets.data(), b->W);
        if (rc == SONIC_OK)
            rc = sonic_prefill_enqueue(h, b->has_req_win ? b->req_win.data() : nullptr, b->R, b->prompt_ids.data(), b->prompt_off.data(), b->max_new.data());
        std::unique_lock<std::mutex> lk(p->mu);
        if (rc != SONIC_OK) { finish(p, b, rc, sonic_last_error(h)); continue; }      // a bad request fails alone
        auto r = std::make_shared<Ready>(); r->b = b; r->src = h;
        p->ready.push_back(r);
        p->cv.notify_all();
        const auto tw0 = std::chrono::steady_clock::now();
        p->cv.wait(lk, [&] { return r->taken || p->failed; });     // the handle's rows are the splice's source until a decoder has queued the copy
        p->prefill_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        if (!r->taken) {                                            // the decode side died with this batch still in hand
            for (auto it = p->ready.begin(); it != p->ready.end(); ++it) if (*it == r) { p->ready.erase(it); break; }
            finish(p, b, p->failed, p->fail_msg);
        }
    }
}

void decode_thread(sonic_pipeline* p, int k) {
    sonic_engine* d = p->dec[k];
    const int nb = p->blocks_per_dec, B = p->block;
    struct Occ { std::shared_ptr<Batch> b; int64_t valid_after = 0; };
    std::vector<Occ> occ(nb);
    int n_occ = 0;
    int32_t fin[64], nn[64];
    auto fail_all = [&](int rc) {
        std::unique_lock<std::mutex> lk(p->mu);
        if (!p->failed) { p->failed = rc ? rc : SONIC_ERR_HIP; p->fail_msg = sonic_last_error(d); }
        for (auto& o : occ) if (o.b) { finish(p, o.b, p->failed, p->fail_msg); o.b.reset(); }
        p->cv.notify_all();
    };
    for (;;) {
        // ---- take hand-overs into free blocks
        {
            std::unique_lock<std::mutex> lk(p->mu);
            for (;;) {
                if (p->failed) { lk.unlock(); fail_all(p->failed); return; }
                p->half[k] = n_occ > 0 && n_occ < nb;
                bool other_half = false;
                for (size_t j = 0; j < p->half.size(); ++j) if ((int)j != k && p->half[j]) other_half = true;
                const bool may_take = n_occ < nb && !p->ready.empty() && !(p->pair && n_occ == 0 && other_half);
                if (may_take) {
                    auto r = p->ready.front(); p->ready.pop_front();
                    int blk = 0; while (occ[blk].b) ++blk;            // lowest free block: the loop steps only as many rows as are occupied
                    std::vector<int32_t> src(r->b->R), dst(r->b->R);
                    for (int i = 0; i < r->b->R; ++i) { src[i] = i; dst[i] = blk * B + i; }
                    lk.unlock();
                    int64_t seq = 0;
                    const int rc = sonic_splice_rows(d, r->src, r->b->R, src.data(), dst.data(), &seq);
                    lk.lock();
                    if (rc != SONIC_OK) {                              // (the source handle is fine; this decoder is not)
                        p->ready.push_front(r);
                        lk.unlock(); fail_all(rc); return;
                    }
                    occ[blk].b = r->b; occ[blk].valid_after = seq; ++n_occ;
                    r->taken = true;
                    p->cv.notify_all();
                    continue;
                }
                if (n_occ > 0) break;                                 // rows are running: go and step them
                if (p->stop && p->queue.empty() && p->ready.empty()) {
                    // nothing running here, nothing to come?  Batches may still be inside a prefill thread: they show up in `all` as running
                    bool pending = false;
                    for (auto& b : p->all) if (b->status == -1) pending = true;
                    if (!pending) return;
                }
                const auto ti0 = std::chrono::steady_clock::now();
                p->cv.wait(lk);
                p->decoder_idle_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count();
            }
        }
        // ---- one chunk for the occupied rows; a second one before fetching when other blocks keep running
        int top = 0;
        for (int i = 0; i < nb; ++i) if (occ[i].b) top = (i + 1) * B;
        int64_t seq = 0; int32_t nact = 0;
        int rc = sonic_service_step(d, 1, top, fin, nn, &seq, &nact);
        if (rc != SONIC_OK) { fail_all(rc); return; }
        int steps = 1;
        std::vector<int> done;
        for (int i = 0; i < nb; ++i) {
            if (!occ[i].b || seq <= occ[i].valid_after) continue;
            bool all_fin = true;
            for (int r = 0; r < occ[i].b->R; ++r) all_fin = all_fin && fin[i * B + r];
            if (all_fin) done.push_back(i);
        }
        if (!done.empty() && (int)done.size() < n_occ) {
            // fetching a block's rows takes this thread about as long as a chunk takes the device: the other block's next chunk goes out first
            int32_t f2[64], n2[64]; int64_t s2 = 0;
            rc = sonic_service_step(d, 1, top, f2, n2, &s2, &nact);
            if (rc != SONIC_OK) { fail_all(rc); return; }
            ++steps;
        }
        for (int i : done) {
            Batch& b = *occ[i].b;
            std::vector<int32_t> rows(b.R), counts(b.R);
            for (int r = 0; r < b.R; ++r) { rows[r] = i * B + r; counts[r] = nn[i * B + r]; if (counts[r] > b.out_ld) counts[r] = b.out_ld; }
            rc = sonic_fetch_rows(d, b.R, rows.data(), counts.data(), b.out_ids, b.out_ld);
            if (rc != SONIC_OK) { fail_all(rc); return; }
            for (int r = 0; r < b.R; ++r) b.out_len[r] = counts[r];
            std::unique_lock<std::mutex> lk(p->mu);
            finish(p, occ[i].b, SONIC_OK, "");
            occ[i].b.reset(); --n_occ;
        }
        std::unique_lock<std::mutex> lk(p->mu);
        p->chunks += steps;
        (top == nb * B ? p->chunks_full : p->chunks_part) += steps;
    }
}

}   // namespace

extern "C" {

SONIC_API int sonic_pipeline_create(sonic_engine* const* decoders, int n_dec, sonic_engine* const* prefills, int n_pre, int block, int rows_per_decoder,
                                    sonic_pipeline** out) {
    if (!out) return SONIC_ERR_INVALID;
    *out = nullptr;
    if (!decoders || !prefills || n_dec < 1 || n_pre < 1 || block < 1 || block > 64 || rows_per_decoder < block || rows_per_decoder > 64) return SONIC_ERR_INVALID;
    // every handle once, all on one weight copy and one device; a decoding handle holds rows_per_decoder rows, a prefill handle one block
    {
        const void* w0 = nullptr; int32_t dev0 = -1;
        for (int i = 0; i < n_dec + n_pre; ++i) {
            sonic_engine* h = i < n_dec ? decoders[i] : prefills[i - n_dec];
            int32_t mb = 0, dev = 0; const void* wid = nullptr;
            if (!h || sonic_engine_info(h, &mb, nullptr, nullptr, &dev, &wid) != SONIC_OK) return SONIC_ERR_INVALID;
            for (int j = 0; j < i; ++j) if (h == (j < n_dec ? decoders[j] : prefills[j - n_dec])) return SONIC_ERR_INVALID;
            if (i == 0) { w0 = wid; dev0 = dev; }
            if (wid != w0 || dev != dev0) return SONIC_ERR_INVALID;
            if (mb < (i < n_dec ? rows_per_decoder : block)) return SONIC_ERR_INVALID;
        }
    }
    sonic_pipeline* p = new sonic_pipeline();
    p->dec.assign(decoders, decoders + n_dec); p->pre.assign(prefills, prefills + n_pre);
    p->block = block; p->blocks_per_dec = rows_per_decoder / block;
    p->half.assign(n_dec, 0);
    for (int i = 0; i < n_dec; ++i) {
        const int rc = sonic_service_begin(p->dec[i]);
        if (rc != SONIC_OK) {
            p->last_err = sonic_last_error(p->dec[i]);
            for (int j = 0; j < i; ++j) (void)sonic_service_end(p->dec[j]);
            delete p;
            return rc;
        }
    }
    p->began = true;
    for (int i = 0; i < n_pre; ++i) p->threads.emplace_back(prefill_thread, p, p->pre[i]);
    for (int i = 0; i < n_dec; ++i) p->threads.emplace_back(decode_thread, p, i);
    *out = p;
    return SONIC_OK;
}

SONIC_API int sonic_pipeline_submit(sonic_pipeline* p, const int16_t* pcm, const int64_t* offsets, int W, const int32_t* req_win, int R,
                                    const int32_t* prompt_ids, const int64_t* prompt_off, const int32_t* max_new,
                                    int32_t* out_ids, int out_ld, int32_t* out_len, int64_t* ticket_out) {
    if (!p || !prompt_ids || !prompt_off || !max_new || !out_ids || !out_len || R < 1 || R > p->block || out_ld < 1) return SONIC_ERR_INVALID;
    if (pcm && (!offsets || W < 1)) return SONIC_ERR_INVALID;
    auto b = std::make_shared<Batch>();
    b->pcm = pcm; b->W = W;
    if (pcm) b->offsets.assign(offsets, offsets + W + 1);
    if (req_win) { b->req_win.assign(req_win, req_win + R + 1); b->has_req_win = true; }
    b->R = R;
    b->prompt_off.assign(prompt_off, prompt_off + R + 1);
    b->prompt_ids.assign(prompt_ids, prompt_ids + prompt_off[R]);
    b->max_new.assign(max_new, max_new + R);
    b->out_ids = out_ids; b->out_ld = out_ld; b->out_len = out_len;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->stop) { p->last_err = "pipeline is closed"; return SONIC_ERR_INVALID; }
    if (p->failed) { p->last_err = p->fail_msg; return p->failed; }
    b->ticket = p->next_ticket++;
    p->queue.push_back(b); p->all.push_back(b);
    if (ticket_out) *ticket_out = b->ticket;
    p->cv.notify_all();
    return SONIC_OK;
}

// ticket > 0: blocks until that batch is complete and returns ITS status (then forgets it); ticket 0: until every batch submitted so far is
// complete, returns the first failure among them (or SONIC_OK) and forgets them all
SONIC_API int sonic_pipeline_wait(sonic_pipeline* p, int64_t ticket) {
    if (!p) return SONIC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(p->mu);
    if (ticket > 0) {
        std::shared_ptr<Batch> b;
        for (auto& x : p->all) if (x->ticket == ticket) b = x;
        if (!b) { p->last_err = "unknown ticket"; return SONIC_ERR_INVALID; }
        p->cv.wait(lk, [&] { return b->status != -1; });
        for (auto it = p->all.begin(); it != p->all.end(); ++it) if (*it == b) { p->all.erase(it); break; }
        if (b->status != SONIC_OK) p->last_err = b->err;
        return b->status;
    }
    const int64_t upto = p->next_ticket;
    p->cv.wait(lk, [&] { for (auto& x : p->all) if (x->ticket < upto && x->status == -1) return false; return true; });
    int rc = SONIC_OK;
    for (auto it = p->all.begin(); it != p->all.end();) {
        if ((*it)->ticket < upto) { if (rc == SONIC_OK && (*it)->status != SONIC_OK) { rc = (*it)->status; p->last_err = (*it)->err; } it = p->all.erase(it); }
        else ++it;
    }
    return rc;
}

SONIC_API int sonic_pipeline_stats(sonic_pipeline* p, int64_t* batches_done, int64_t* chunks_queued, int32_t* batches_in_flight_max) {
    if (!p) return SONIC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(p->mu);
    if (batches_done) *batches_done = p->done_batches;
    if (chunks_queued) *chunks_queued = p->chunks;
    if (batches_in_flight_max) *batches_in_flight_max = (int32_t)(p->dec.size() * p->blocks_per_dec + p->pre.size());
    return SONIC_OK;
}

// the message is copied under the lock into a buffer of the calling thread (other threads go on writing p->last_err)
SONIC_API const char* sonic_pipeline_last_error(sonic_pipeline* p) {
    static thread_local std::string mine;
    if (!p) return "";
    std::unique_lock<std::mutex> lk(p->mu);
    mine = p->last_err;
    return mine.c_str();
}

// waits for what was submitted, stops the threads, takes the decoding handles out of continuous mode (the handles themselves stay the caller's)
SONIC_API int sonic_pipeline_destroy(sonic_pipeline* p) {
    if (!p) return SONIC_OK;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->stop = true;
        p->cv.notify_all();
    }
    for (auto& t : p->threads) t.join();
    if (p->began) for (auto* d : p->dec) (void)sonic_service_end(d);
    if (getenv("SONIC_PIPE_STATS"))
        fprintf(stderr, "[sonic] pipeline: %lld batches, %lld chunks (%lld with every block of the loop occupied, %lld part-filled); prefill threads waited %.0f ms for a "
                        "free block, decoding threads idled %.0f ms with nothing to step\n", (long long)p->done_batches, (long long)p->chunks, (long long)p->chunks_full,
                (long long)p->chunks_part, p->prefill_wait_ms, p->decoder_idle_ms);
    delete p;
    return SONIC_OK;
}

}   // extern "C"
