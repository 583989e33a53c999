// sonic_dispatch_*: the request-level scheduler for live traffic as native threads.
//
// What it replaces: sonicscribe_amd/dispatch.py _ContinuousReplica - Python threads over ctypes that prefill whatever is queued, splice the rows into
// a continuously decoding handle, step it chunk by chunk and fetch rows as they finish.  The arithmetic was never in Python, but every decode chunk
// of every loop went through the interpreter (a numpy record, list comprehensions, a condition variable under the GIL) beside the sessions' own
// Python work.  The call sites this serves in the reference: one `await transcribe()` per partial / final of every WebSocket session on the event
// loop (backend/connection_manager.py:127-245, backend/transcription_manager.py:19-65) and three executor threads in file mode
// (backend/main.py:429-445, 616-624) - all serialised on one model object there.
//
// Same schedule as the Python class (whose CPU tests keep describing it, tests/test_dispatch.py):
//   * submit() queues a request (W windows: host PCM or slices of device rings, a prompt, a budget) and returns a ticket;
//   * one thread per PREFILL handle: takes the oldest requests that fit - as many as the emptiest decoder has free rows and the handle has windows,
//     step classes mix - stages them (sonic_stage_mixed), runs log-mel + encoder + prompt forward + first token (sonic_prefill, waited for: a splice
//     queued behind a running prefill would hold the decoder's stream) and hands the batch over; a request that fails validation fails alone
//     (the batch is retried one request at a time);
//   * one thread per DECODING handle: splices handed-over rows into its lowest free rows between two chunks, queues chunks over the occupied
//     rows (sonic_service_step), fetches every row the moment the pipelined check shows it finished (sonic_fetch_rows) and completes its ticket;
//   * completions are collected by sonic_dispatch_next (blocking, from any thread): the host side keeps one thread that turns them into futures.
// No thread polls: condition variables here, blocking HIP events inside the engine.  Only the C ABI of include/sonic_hip.h is used.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
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

struct Req {
    int64_t ticket = 0;
    int W = 0;
    std::vector<int16_t> pcm; std::vector<int64_t> off;            // host windows (ring windows: empty ranges)
    std::vector<sonic_ring*> rings; std::vector<int64_t> ring_start; std::vector<int32_t> ring_n; bool any_ring = false;
    std::vector<int32_t> prompt; int max_new = 0;
    int status = -1; std::string err; std::vector<int32_t> ids;    // result
    bool cancelled = false;
};
typedef std::shared_ptr<Req> ReqP;
struct Handover { std::vector<ReqP> reqs; sonic_engine* src = nullptr; bool taken = false; };

}   // namespace

struct sonic_dispatch {
    std::vector<sonic_engine*> dec, pre;
    int n_rows = 0, pre_cap = 0;
    bool adaptive_tiles = true;
    std::mutex mu;
    std::condition_variable cv;                                    // queue, hand-overs, free rows, state
    std::condition_variable cv_done;                               // completions
    std::deque<ReqP> q;
    std::vector<int> free_rows;                                    // per decoder: rows neither occupied nor reserved by a prefill in flight
    std::vector<std::deque<std::shared_ptr<Handover>>> hand;
    std::deque<ReqP> done;
    int64_t next_ticket = 1, batches = 0, chunks = 0, outstanding = 0;   // outstanding: submitted and not yet collected by sonic_dispatch_next
    int queued_windows = 0;
    bool stop = false;
    int failed = 0; std::string fail_msg;
    std::vector<std::thread> threads;
};

namespace {

void finish(sonic_dispatch* d, const ReqP& r, int status, const std::string& err) {   // d->mu held
    if (r->status != -1) return;
    r->status = status; r->err = err;
    d->done.push_back(r);
    d->cv_done.notify_all();
}

// stage + prefill `batch` on handle h; SONIC_OK or the failing status (message in *err)
int prefill_batch(sonic_dispatch* d, sonic_engine* h, const std::vector<ReqP>& batch, bool busy, std::string* err) {
    std::vector<int16_t> pcm; std::vector<int64_t> off{0}; std::vector<sonic_ring*> rings; std::vector<int64_t> rstart; std::vector<int32_t> rn;
    std::vector<int32_t> req_win{0}, ids, mn; std::vector<int64_t> poff{0};
    bool any_ring = false;
    for (auto& r : batch) {
        for (int w = 0; w < r->W; ++w) {
            const int64_t a = r->off[w], b = r->off[w + 1];
            pcm.insert(pcm.end(), r->pcm.begin() + a, r->pcm.begin() + b);
            off.push_back((int64_t)pcm.size());
            rings.push_back(r->any_ring ? r->rings[w] : nullptr); rstart.push_back(r->any_ring ? r->ring_start[w] : 0); rn.push_back(r->any_ring ? r->ring_n[w] : 0);
            any_ring = any_ring || (r->any_ring && r->rings[w]);
        }
        req_win.push_back((int32_t)(off.size() - 1));
        ids.insert(ids.end(), r->prompt.begin(), r->prompt.end());
        poff.push_back((int64_t)ids.size());
        mn.push_back(r->max_new);
    }
    if (pcm.empty()) pcm.push_back(0);
    const int W = (int)off.size() - 1, R = (int)batch.size();
    // A prefill beside RUNNING rows keeps the big GEMM tiles even where they under-fill the chip: the idle CUs are where the decode loops' kernels
    // run meanwhile (profiles/round4_streaming_ab.txt).  The tile choice never changes a result's bits.
    if (d->adaptive_tiles) (void)sonic_set_option(h, "gemm_small_eff", busy ? 0 : 75);
    int rc = any_ring ? sonic_stage_mixed(h, pcm.data(), off.data(), rings.data(), rstart.data(), rn.data(), W, req_win.data(), R)
                      : sonic_stage_pcm(h, pcm.data(), off.data(), W);
    if (rc == SONIC_OK) rc = sonic_prefill(h, req_win.data(), R, ids.data(), poff.data(), mn.data(), 0);
    if (rc != SONIC_OK && err) *err = sonic_last_error(h);
    return rc;
}

void release_rows(sonic_dispatch* d, int k, int n) {             // d->mu held
    d->free_rows[k] += n;
    d->cv.notify_all();
}

// hands a prefilled batch to decoder k and waits until its rows have been spliced (they are the splice's source until then)
void hand_over(sonic_dispatch* d, sonic_engine* h, std::vector<ReqP> batch, int k, std::unique_lock<std::mutex>& lk) {
    auto ho = std::make_shared<Handover>(); ho->reqs = std::move(batch); ho->src = h;
    d->hand[k].push_back(ho);
    d->cv.notify_all();
    d->cv.wait(lk, [&] { return ho->taken || d->failed; });
    if (!ho->taken) {                                              // the decode side died with this batch in hand: fail it here, exactly once
        auto& v = d->hand[k];
        auto it = std::find(v.begin(), v.end(), ho);
        if (it != v.end()) {
            v.erase(it);
            for (auto& r : ho->reqs) finish(d, r, d->failed, d->fail_msg);
            release_rows(d, k, (int)ho->reqs.size());
        }
    }
}

void prefill_thread(sonic_dispatch* d, sonic_engine* h) {
    const int nd = (int)d->dec.size();
    for (;;) {
        std::vector<ReqP> batch; int k = 0; bool busy = false;
        {
            std::unique_lock<std::mutex> lk(d->mu);
            d->cv.wait(lk, [&] { return d->stop || d->failed || (!d->q.empty() && *std::max_element(d->free_rows.begin(), d->free_rows.end()) > 0); });
            if (d->stop || d->failed) return;                      // closing / failed: nothing new is started (destroy / the decode side fails what is queued)
            k = (int)(std::max_element(d->free_rows.begin(), d->free_rows.end()) - d->free_rows.begin());   // the emptiest decoder takes the whole batch
            int used = 0;
            while (!d->q.empty()) {                                // oldest first, whatever fits the handle's windows and the free rows; classes mix
                ReqP r = d->q.front();
                if (r->cancelled) { d->q.pop_front(); d->queued_windows -= r->W; finish(d, r, SONIC_ERR_INVALID, "cancelled"); continue; }
                if (r->W > d->pre_cap) {
                    d->q.pop_front(); d->queued_windows -= r->W;
                    char m[128]; snprintf(m, sizeof m, "audio spans %d windows, engine max_batch is %d", r->W, d->pre_cap);
                    finish(d, r, SONIC_ERR_INVALID, m);
                    continue;
                }
                if ((int)batch.size() >= d->free_rows[k] || used + r->W > d->pre_cap) break;
                d->q.pop_front(); d->queued_windows -= r->W;
                batch.push_back(r); used += r->W;
            }
            if (batch.empty()) continue;
            d->free_rows[k] -= (int)batch.size();                  // reserved until the rows are fetched (or the prefill fails)
            ++d->batches;
            int running = 0; for (int j = 0; j < nd; ++j) running += d->n_rows - d->free_rows[j];
            busy = running > (int)batch.size();                    // rows running or reserved besides this batch's own
        }
        std::string err;
        int rc = prefill_batch(d, h, batch, busy, &err);
        std::unique_lock<std::mutex> lk(d->mu);
        if (rc == SONIC_OK) { hand_over(d, h, std::move(batch), k, lk); continue; }
        if (batch.size() == 1) { finish(d, batch[0], rc, err); release_rows(d, k, 1); continue; }
        // a per-request validation error must not poison its neighbours: one by one
        for (auto& r : batch) {
            lk.unlock();
            std::string e1;
            const int rc1 = prefill_batch(d, h, std::vector<ReqP>{r}, busy, &e1);
            lk.lock();
            if (rc1 != SONIC_OK) { finish(d, r, rc1, e1); release_rows(d, k, 1); }
            else hand_over(d, h, std::vector<ReqP>{r}, k, lk);
        }
    }
}

void decode_thread(sonic_dispatch* d, int k) {
    sonic_engine* e = d->dec[k];
    const int n_rows = d->n_rows;
    std::vector<ReqP> rows(n_rows);
    std::vector<int64_t> valid_after(n_rows, 0);
    int occupied = 0;
    int32_t fin[64], nn[64];
    auto fail_all = [&](int rc) {                                  // the engine failed: nothing queued or in flight on this loop can complete
        std::unique_lock<std::mutex> lk(d->mu);
        if (!d->failed) { d->failed = rc ? rc : SONIC_ERR_HIP; d->fail_msg = sonic_last_error(e); }
        for (auto& r : rows) if (r) { finish(d, r, d->failed, d->fail_msg); r.reset(); }
        for (auto& ho : d->hand[k]) { for (auto& r : ho->reqs) finish(d, r, d->failed, d->fail_msg); ho->taken = true; }
        d->hand[k].clear();
        while (!d->q.empty()) { finish(d, d->q.front(), d->failed, d->fail_msg); d->q.pop_front(); }
        d->queued_windows = 0;
        d->cv.notify_all();
    };
    for (;;) {
        std::vector<std::shared_ptr<Handover>> hs;
        {
            std::unique_lock<std::mutex> lk(d->mu);
            // idle = no row occupied and none reserved by a prefill in flight (free_rows counts both): a closing dispatcher keeps its decode loops
            // until every prefill that has taken rows has handed them over and they are fetched
            d->cv.wait(lk, [&] { return !d->hand[k].empty() || occupied > 0 || d->failed || (d->stop && d->free_rows[k] == n_rows); });
            if (d->failed) { lk.unlock(); fail_all(d->failed); return; }
            hs.assign(d->hand[k].begin(), d->hand[k].end()); d->hand[k].clear();
            if (d->stop && hs.empty() && occupied == 0 && d->free_rows[k] == n_rows) return;
        }
        for (auto& ho : hs) {
            const int n = (int)ho->reqs.size();
            std::vector<int32_t> src(n), dst;
            for (int i = 0; i < n_rows && (int)dst.size() < n; ++i) if (!rows[i]) dst.push_back(i);      // lowest free rows: a light load stays in the first 16
            for (int i = 0; i < n; ++i) src[i] = i;
            int64_t seq = 0;
            const int rc = (int)dst.size() == n ? sonic_splice_rows(e, ho->src, n, src.data(), dst.data(), &seq) : SONIC_ERR_INVALID;
            if (rc != SONIC_OK) {
                { std::unique_lock<std::mutex> lk(d->mu); d->hand[k].push_front(ho); }
                fail_all(rc); return;
            }
            for (int i = 0; i < n; ++i) { rows[dst[i]] = ho->reqs[i]; valid_after[dst[i]] = seq; }
            occupied += n;
            std::unique_lock<std::mutex> lk(d->mu);
            ho->taken = true;
            d->cv.notify_all();
        }
        if (occupied == 0) continue;
        int top = 0;
        for (int i = 0; i < n_rows; ++i) if (rows[i]) top = i + 1;
        int64_t seq = 0; int32_t nact = 0;
        int rc = sonic_service_step(e, 1, top, fin, nn, &seq, &nact);
        if (rc != SONIC_OK) { fail_all(rc); return; }
        std::vector<int32_t> dr, dc;
        for (int i = 0; i < n_rows; ++i) if (rows[i] && seq > valid_after[i] && fin[i]) { dr.push_back(i); dc.push_back(nn[i]); }
        int ld = 1; for (int c : dc) ld = c > ld ? c : ld;
        std::vector<int32_t> out;
        if (!dr.empty()) {                                         // one call for all of them: one wait, one release launch
            out.assign((size_t)dr.size() * ld, 0);
            rc = sonic_fetch_rows(e, (int)dr.size(), dr.data(), dc.data(), out.data(), ld);
            if (rc != SONIC_OK) { fail_all(rc); return; }
        }
        std::unique_lock<std::mutex> lk(d->mu);
        ++d->chunks;
        for (size_t j = 0; j < dr.size(); ++j) {
            ReqP r = rows[dr[j]]; rows[dr[j]].reset(); --occupied;
            r->ids.assign(out.begin() + j * ld, out.begin() + j * ld + dc[j]);
            finish(d, r, SONIC_OK, "");
        }
        if (!dr.empty()) release_rows(d, k, (int)dr.size());
    }
}

}   // namespace

extern "C" {

SONIC_API int sonic_dispatch_create(sonic_engine* const* decoders, int n_dec, sonic_engine* const* prefills, int n_pre, int adaptive_tiles, sonic_dispatch** out) {
    if (!out) return SONIC_ERR_INVALID;
    *out = nullptr;
    if (!decoders || !prefills || n_dec < 1 || n_pre < 1) return SONIC_ERR_INVALID;
    int32_t rows = 0, cap = 0;
    {   // every handle once, all on one weight copy and one device
        const void* w0 = nullptr; int32_t dev0 = -1;
        for (int i = 0; i < n_dec + n_pre; ++i) {
            sonic_engine* h = i < n_dec ? decoders[i] : prefills[i - n_dec];
            int32_t mb = 0, dev = 0; const void* wid = nullptr;
            if (!h || sonic_engine_info(h, &mb, nullptr, nullptr, &dev, &wid) != SONIC_OK) return SONIC_ERR_INVALID;
            for (int j = 0; j < i; ++j) if (h == (j < n_dec ? decoders[j] : prefills[j - n_dec])) return SONIC_ERR_INVALID;
            if (i == 0) { w0 = wid; dev0 = dev; rows = mb; }
            if (wid != w0 || dev != dev0) return SONIC_ERR_INVALID;
            if (i < n_dec) rows = mb < rows ? mb : rows; else cap = (i == n_dec || mb < cap) ? mb : cap;
        }
    }
    if (rows < 1 || rows > 64 || cap < 1) return SONIC_ERR_INVALID;
    sonic_dispatch* d = new sonic_dispatch();
    d->dec.assign(decoders, decoders + n_dec); d->pre.assign(prefills, prefills + n_pre);
    d->n_rows = rows; d->pre_cap = cap; d->adaptive_tiles = adaptive_tiles != 0;
    d->free_rows.assign(n_dec, rows); d->hand.resize(n_dec);
    for (int i = 0; i < n_dec; ++i) {
        const int rc = sonic_service_begin(d->dec[i]);
        if (rc != SONIC_OK) { for (int j = 0; j < i; ++j) (void)sonic_service_end(d->dec[j]); delete d; return rc; }
    }
    for (int i = 0; i < n_dec; ++i) d->threads.emplace_back(decode_thread, d, i);
    for (int i = 0; i < n_pre; ++i) d->threads.emplace_back(prefill_thread, d, d->pre[i]);
    *out = d;
    return SONIC_OK;
}

// One request of W windows.  Window w: host samples host_pcm[host_off[w] .. host_off[w + 1]) (int16, already peak-normalised over the request, as
// sonic_stage_pcm takes them) when rings is NULL or rings[w] is NULL, else samples [ring_start[w], ring_start[w] + ring_n[w]) of rings[w] (raw wire
// PCM, normalised on the device over the request's windows).  Everything is copied before the call returns.
SONIC_API int sonic_dispatch_submit(sonic_dispatch* d, const int16_t* host_pcm, const int64_t* host_off, sonic_ring* const* rings, const int64_t* ring_start,
                                    const int32_t* ring_n, int W, const int32_t* prompt_ids, int prompt_len, int max_new, int64_t* ticket_out) {
    if (!d || !ticket_out || W < 1 || !prompt_ids || prompt_len < 1 || max_new < 1 || !host_off) return SONIC_ERR_INVALID;
    if (rings && (!ring_start || !ring_n)) return SONIC_ERR_INVALID;
    auto r = std::make_shared<Req>();
    r->W = W;
    r->off.assign(host_off, host_off + W + 1);
    if (r->off[0] != 0) return SONIC_ERR_INVALID;
    if (r->off[W] > 0) { if (!host_pcm) return SONIC_ERR_INVALID; r->pcm.assign(host_pcm, host_pcm + r->off[W]); }
    if (rings) {
        r->rings.assign(rings, rings + W); r->ring_start.assign(ring_start, ring_start + W); r->ring_n.assign(ring_n, ring_n + W);
        for (int w = 0; w < W; ++w) r->any_ring = r->any_ring || rings[w];
    }
    r->prompt.assign(prompt_ids, prompt_ids + prompt_len); r->max_new = max_new;
    std::unique_lock<std::mutex> lk(d->mu);
    if (d->stop) return SONIC_ERR_INVALID;
    if (d->failed) return d->failed;
    r->ticket = d->next_ticket++;
    d->q.push_back(r); d->queued_windows += W; ++d->outstanding;
    *ticket_out = r->ticket;
    d->cv.notify_all();
    return SONIC_OK;
}

// a request that is still queued leaves the queue (it completes with status SONIC_ERR_INVALID, "cancelled"); one that has reached a handle runs on
SONIC_API int sonic_dispatch_cancel(sonic_dispatch* d, int64_t ticket) {
    if (!d) return SONIC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(d->mu);
    for (auto it = d->q.begin(); it != d->q.end(); ++it)
        if ((*it)->ticket == ticket) { ReqP r = *it; d->q.erase(it); d->queued_windows -= r->W; r->cancelled = true; finish(d, r, SONIC_ERR_INVALID, "cancelled"); return SONIC_OK; }
    return SONIC_ERR_INVALID;
}

// The next completed request, in completion order.  Blocks up to timeout_ms (< 0: until one completes, or the dispatcher has been closed and every
// request has been collected).  *ticket_out = 0: none (timeout, or closed and drained).  Otherwise *status_out is the request's sonic_status, its *n_out tokens are in out_ids
// (at most out_cap are copied) and err (if given) holds the engine's message for a failed request.
SONIC_API int sonic_dispatch_next(sonic_dispatch* d, int timeout_ms, int64_t* ticket_out, int32_t* status_out, int32_t* out_ids, int out_cap, int32_t* n_out,
                                  char* err, int err_cap) {
    if (!d || !ticket_out) return SONIC_ERR_INVALID;
    *ticket_out = 0;
    std::unique_lock<std::mutex> lk(d->mu);
    auto ready = [&] { return !d->done.empty() || (d->stop && d->outstanding == 0); };
    if (timeout_ms < 0) d->cv_done.wait(lk, ready);
    else if (!d->cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) return SONIC_OK;
    if (d->done.empty()) return SONIC_OK;
    ReqP r = d->done.front(); d->done.pop_front(); --d->outstanding;
    *ticket_out = r->ticket;
    if (status_out) *status_out = r->status;
    const int n = (int)r->ids.size();
    if (n_out) *n_out = n;
    if (out_ids) memcpy(out_ids, r->ids.data(), (size_t)(n < out_cap ? n : out_cap) * 4);
    if (err && err_cap > 0) { strncpy(err, r->err.c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
    if (d->stop && d->outstanding == 0) d->cv_done.notify_all();
    return SONIC_OK;
}

SONIC_API int sonic_dispatch_stats(sonic_dispatch* d, int64_t* prefill_batches, int64_t* decode_chunks, int32_t* load_windows, int32_t* free_rows) {
    if (!d) return SONIC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(d->mu);
    int fr = 0; for (int f : d->free_rows) fr += f;
    if (prefill_batches) *prefill_batches = d->batches;
    if (decode_chunks) *decode_chunks = d->chunks;
    if (load_windows) *load_windows = (int32_t)(d->n_rows * (int)d->dec.size() - fr + d->queued_windows);   // rows occupied or reserved + windows queued
    if (free_rows) *free_rows = fr;
    return SONIC_OK;
}

// requests still queued fail at once ("ASR engine is closed"), requests that have reached a handle complete; their results stay collectable by
// sonic_dispatch_next until sonic_dispatch_destroy.  Idempotent.
SONIC_API int sonic_dispatch_close(sonic_dispatch* d) {
    if (!d) return SONIC_OK;
    {
        std::unique_lock<std::mutex> lk(d->mu);
        if (d->stop) return SONIC_OK;
        d->stop = true;
        while (!d->q.empty()) { finish(d, d->q.front(), SONIC_ERR_INVALID, "ASR engine is closed"); d->q.pop_front(); }
        d->queued_windows = 0;
        d->cv.notify_all();
    }
    for (auto& t : d->threads) t.join();
    d->threads.clear();
    for (auto* e : d->dec) (void)sonic_service_end(e);
    std::unique_lock<std::mutex> lk(d->mu);
    d->cv_done.notify_all();
    return SONIC_OK;
}

SONIC_API int sonic_dispatch_destroy(sonic_dispatch* d) {
    if (!d) return SONIC_OK;
    (void)sonic_dispatch_close(d);
    delete d;
    return SONIC_OK;
}

}   // extern "C"
