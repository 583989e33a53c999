"""Host-side dispatcher (sonicscribe_amd/dispatch.py) on CPU with stub engines: placement over replicas, step-class buckets,
no linger on an idle replica, error isolation, and the asyncio entry the WebSocket callers use
(backend/transcription_manager.py:43-65, backend/connection_manager.py:127-245)."""
import asyncio
import threading
import time

import numpy as np
import pytest

from sonicscribe_amd.dispatch import Dispatcher, step_class


class StubEngine:
    """Duck-typed engine: records its batches, 'decodes' a request to [sum(pcm) % 1000, len(prompt), max_new]."""

    def __init__(self, max_batch=4, delay=0.0):
        self.max_batch, self.delay = max_batch, delay
        self.batches = []
        self.lock = threading.Lock()
        self.gate = None

    def transcribe_batch(self, segs, prompts, max_new, req_win=None):
        if self.gate is not None:
            self.gate.wait()
        time.sleep(self.delay)
        with self.lock:
            self.batches.append((len(segs), list(max_new)))
        out = []
        for r in range(len(prompts)):
            w = segs[req_win[r]:req_win[r + 1]]
            if any(len(x) == 13 for x in w):
                raise ValueError("Audio features and audio tokens do not match")
            out.append(np.asarray([int(sum(int(x.sum()) for x in w)) % 1000, len(prompts[r]), max_new[r]], np.int32))
        return out, None


def seg(v, n=16):
    return np.full(n, v, np.int16)


def test_step_classes():
    assert step_class(15) == 0 and step_class(16) == 0 and step_class(17) == 1 and step_class(150) == 2 and step_class(256) == 2 and step_class(300) == 3


def test_results_map_back_and_batches_form_under_load():
    e = StubEngine(max_batch=4)
    e.gate = threading.Event()                         # hold the first batch on the "device" while the others queue up
    d = Dispatcher([e])
    futs = [d.submit([seg(0)], [1, 2, 3], 150)]
    time.sleep(0.05)                                   # the worker has taken it and sits on the "device"
    futs += [d.submit([seg(i)], [1, 2, 3], 150) for i in range(1, 9)]
    time.sleep(0.02)
    e.gate.set()
    res = [f.result(timeout=5) for f in futs]
    assert [int(r[0]) for r in res] == [(16 * i) % 1000 for i in range(9)]
    assert e.batches[0][0] == 1                        # the first request started alone: nothing to wait for
    assert sorted(b[0] for b in e.batches[1:]) == [4, 4]   # the queued ones were batched up to max_batch
    d.close()


def test_no_linger_on_idle_replica():
    e = StubEngine()
    d = Dispatcher([e])
    t0 = time.perf_counter()
    d.submit([seg(1)], [1], 15).result(timeout=5)
    assert time.perf_counter() - t0 < 0.05            # (the old coalescer added a fixed 2 ms linger; there is none now)
    d.close()


def test_partials_never_ride_finals():
    e = StubEngine(max_batch=8)
    e.gate = threading.Event()
    d = Dispatcher([e])
    first = d.submit([seg(0)], [1], 150)
    time.sleep(0.02)
    futs = [d.submit([seg(i)], [1], 15 if i % 2 else 150) for i in range(1, 9)]
    e.gate.set()
    [f.result(timeout=5) for f in futs + [first]]
    for n, budgets in e.batches:
        assert len({step_class(b) for b in budgets}) == 1, budgets
    d.close()


def test_sessions_stick_and_free_segments_balance():
    engines = [StubEngine(max_batch=4, delay=0.01) for _ in range(4)]
    d = Dispatcher(engines)
    homes = {}
    for rnd in range(3):
        for s in range(16):
            rep = d.pick(f"client-{s}")
            homes.setdefault(s, rep.index)
            assert rep.index == homes[s]               # idle system: a session always lands on its home replica
    assert len(set(homes.values())) > 1
    futs = [d.submit([seg(i)], [1], 150) for i in range(32)]         # keyless: least-loaded placement
    [f.result(timeout=10) for f in futs]
    per = [sum(b[0] for b in e.batches) for e in engines]
    assert sum(per) == 32 and min(per) >= 4, per
    d.close()


def test_overloaded_home_replica_is_rebalanced():
    a, b = StubEngine(max_batch=2), StubEngine(max_batch=2)
    a.gate = threading.Event(); b.gate = threading.Event()
    d = Dispatcher([a, b])
    key = next(k for k in (f"s{i}" for i in range(100)) if d.pick(k).index == 0)
    futs = [d.submit([seg(i)], [1], 150, session=key) for i in range(8)]
    a.gate.set(); b.gate.set()
    [f.result(timeout=5) for f in futs]
    assert sum(x[0] for x in b.batches) >= 1           # the backlog beyond one batch spilled to the other replica
    d.close()


def test_bad_request_does_not_poison_neighbours():
    e = StubEngine(max_batch=4)
    e.gate = threading.Event()
    d = Dispatcher([e])
    f0 = d.submit([seg(0)], [1], 150)
    time.sleep(0.02)
    good = d.submit([seg(3)], [1], 150); bad = d.submit([seg(1, 13)], [1], 150); good2 = d.submit([seg(5)], [1], 150)
    huge = d.submit([seg(1)] * 9, [1], 150)
    e.gate.set()
    assert int(good.result(timeout=5)[0]) == 48 and int(good2.result(timeout=5)[0]) == 80 and f0.result(timeout=5) is not None
    with pytest.raises(ValueError):
        bad.result(timeout=5)
    with pytest.raises(ValueError):
        huge.result(timeout=5)
    d.close()
    with pytest.raises(RuntimeError):
        d.submit([seg(1)], [1], 15)


def test_asyncio_callers_overlap():
    """The WebSocket side: many sessions await their decodes on one event loop; none blocks the loop (the reference blocks it per call)."""
    e = StubEngine(max_batch=8, delay=0.02)
    d = Dispatcher([e])

    async def session(i):
        fut = d.submit([seg(i)], [1], 15, session=f"c{i}")
        return await asyncio.wrap_future(fut)

    async def main():
        t0 = time.perf_counter()
        ticks = 0

        async def heartbeat():
            nonlocal ticks
            while True:
                await asyncio.sleep(0.002); ticks += 1
        hb = asyncio.ensure_future(heartbeat())
        res = await asyncio.gather(*[session(i) for i in range(24)])
        hb.cancel()
        return res, time.perf_counter() - t0, ticks
    res, dt, ticks = asyncio.run(main())
    assert [int(r[0]) for r in res] == [(16 * i) % 1000 for i in range(24)]
    assert dt < 24 * 0.02 * 0.6                         # batched: far less than 24 serial device calls
    assert ticks >= 5                                   # the event loop kept running while the "device" worked
    d.close()


def test_pinned_replica_and_session_home():
    """Requests whose windows live in a device ring are pinned to the ring's replica (Dispatcher.submit(replica=...)); a session's home
    replica is the same crc32 rule pick() uses, so a stream and the session's other decodes land on one GPU."""
    engines = [StubEngine(), StubEngine(), StubEngine()]
    d = Dispatcher(engines)
    homes = {s: d.home(s) for s in ("client-1", "client-2", "client-3", "client-4", "client-5")}
    assert all(0 <= h < 3 for h in homes.values()) and len(set(homes.values())) > 1
    assert homes == {s: d.home(s) for s in homes}                     # stable
    for s, h in homes.items():
        assert d.pick(s).index == h                                   # idle replicas: a session stays at home
    futs = [d.submit([seg(i)], [1, 2, 3], 8, replica=2) for i in range(6)]
    [f.result(timeout=10) for f in futs]
    assert sum(b[0] for b in engines[2].batches) == 6 and not engines[0].batches and not engines[1].batches
    d.close()


def test_cancelled_requests_never_reach_the_engine_and_the_worker_survives():
    """A caller may cancel the future it was handed (a session that disconnected).  The request leaves the queue, its neighbours still
    complete, a cancel that races the batch start is refused, and the replica thread keeps serving afterwards."""
    e = StubEngine(max_batch=4)
    e.gate = threading.Event()
    d = Dispatcher([e])
    first = d.submit([seg(1)], [1], 150)
    time.sleep(0.05)                                   # first is RUNNING on the "device"
    assert not first.cancel()                          # running requests cannot be cancelled
    queued = [d.submit([seg(i)], [1], 150) for i in range(2, 6)]
    assert queued[1].cancel() and queued[2].cancel()   # still pending: cancelled in the queue
    e.gate.set()
    assert int(first.result(timeout=5)[0]) == 16
    assert int(queued[0].result(timeout=5)[0]) == 32 and int(queued[3].result(timeout=5)[0]) == 80
    assert sum(b[0] for b in e.batches) == 3           # the two cancelled requests never reached the engine
    later = d.submit([seg(7)], [1], 150)               # the worker thread is alive
    assert int(later.result(timeout=5)[0]) == 112
    d.close()


def test_text_future_propagates_cancellation_to_the_queued_request():
    from sonicscribe_amd.asr import _text_future
    e = StubEngine(max_batch=1)
    e.gate = threading.Event()
    d = Dispatcher([e])
    d.submit([seg(1)], [1], 15)
    time.sleep(0.05)
    inner = d.submit([seg(2)], [1], 15)
    outer = _text_future(inner, lambda ids: " ".join(map(str, ids)))
    assert outer.cancel() and inner.cancelled()
    e.gate.set()
    ok = _text_future(d.submit([seg(3)], [1, 2], 15), lambda ids: " ".join(str(int(i)) for i in ids))
    assert ok.result(timeout=5) == "48 2 15"
    assert len(e.batches) == 2
    d.close()


# ------------------------------------------------------------------------------------------ continuous (row-level) replica
class StubPool:
    """Duck-typed pair of handles for dispatch._ContinuousReplica: `StubPool.decoder` decodes over n_rows rows (4 steps per chunk), the
    prefill handles compute a request's whole token list up front ([sum(pcm) % 1000, len(prompt)] + range(max_new - 2)) and hand it over row by row."""

    class Prefill:
        def __init__(self, pool, max_batch):
            self.pool, self.max_batch, self.staged, self.rows, self.batches = pool, max_batch, None, [], []
            self.options = []

        def set_option(self, key, value):
            self.options.append((key, value))

        def stage_pcm(self, segs, req_win=None):
            self.staged = (list(segs), list(req_win))

        def prefill(self, prompts, max_new, req_win=None, wait=True):
            segs, rw = self.staged
            rows = []
            for r in range(len(prompts)):
                w = segs[rw[r]:rw[r + 1]]
                if any(len(x) == 13 for x in w):
                    raise ValueError("Audio features and audio tokens do not match")
                toks = ([int(sum(int(x.sum()) for x in w)) % 1000, len(prompts[r])] + list(range(max_new[r])))[:max_new[r]]
                rows.append(toks)
            time.sleep(self.pool.prefill_delay)
            self.rows = rows
            self.batches.append(list(max_new))

    class Decoder:
        def __init__(self, pool, n_rows):
            self.pool, self.max_batch = pool, n_rows
            self.rows = [None] * n_rows                  # [tokens, emitted]
            self.seq, self.on, self.max_occupied = 0, False, 0

        def service_begin(self):
            self.on = True

        def service_end(self):
            self.on = False

        def splice_rows(self, src, src_rows, dst_rows):
            for s, d in zip(src_rows, dst_rows):
                assert self.rows[d] is None, "spliced into an occupied row"
                self.rows[d] = [src.rows[s], 1]
            self.max_occupied = max(self.max_occupied, sum(r is not None for r in self.rows))
            return self.seq

        def service_step(self, n_chunks=1, rows=0):
            assert rows == 0 or all(r is None for r in self.rows[rows:]), "an occupied row lies beyond the rows the caller asked for"
            time.sleep(self.pool.step_delay)
            self.seq += n_chunks
            fin, nn = np.ones(64, np.int32), np.zeros(64, np.int32)
            for i, r in enumerate(self.rows):
                if r is not None:
                    r[1] = min(len(r[0]), r[1] + 4 * n_chunks)
                    fin[i], nn[i] = int(r[1] >= len(r[0])), r[1]
            return fin, nn, self.seq, int(sum(1 for r in self.rows if r is not None and r[1] < len(r[0])))

        def fetch_row(self, row, n):
            toks, self.rows[row] = self.rows[row][0], None
            assert n == len(toks)
            return np.asarray(toks, np.int32)

        def fetch_rows(self, rows, counts):
            return [self.fetch_row(r, n) for r, n in zip(rows, counts)]

    def __init__(self, n_rows=4, prefill_batch=4, n_prefill=1, prefill_delay=0.0, step_delay=0.001):
        self.prefill_delay, self.step_delay = prefill_delay, step_delay
        self.decoder = StubPool.Decoder(self, n_rows)
        self.prefills = [StubPool.Prefill(self, prefill_batch) for _ in range(n_prefill)]


def want_tokens(v, prompt_len, max_new, n=16):
    return ([(n * v) % 1000, prompt_len] + list(range(max_new)))[:max_new]


def test_continuous_replica_rows_join_and_leave_one_by_one():
    pool = StubPool(n_rows=4, prefill_batch=4, prefill_delay=0.002)
    d = Dispatcher([pool.decoder], slots=[pool.prefills], continuous=True)
    budgets = [15, 150, 40, 15, 150, 7, 1, 150, 15, 15, 64, 2]           # partials, finals and a one-token request mixed: no class buckets
    futs = [d.submit([seg(i)], [1, 2, 3], mn) for i, mn in enumerate(budgets)]
    res = [f.result(timeout=10) for f in futs]
    for i, (r, mn) in enumerate(zip(res, budgets)):
        assert r.tolist() == want_tokens(i, 3, mn), i
    assert pool.decoder.max_occupied <= 4 and all(r is None for r in pool.decoder.rows)       # never more rows than the pool has; all handed back
    assert any(len(set(b)) > 1 for b in pool.prefills[0].batches)        # a prefill batch mixed budgets
    # a short request that arrives while long ones decode does not wait for them
    long_f = [d.submit([seg(50 + i)], [1], 150) for i in range(3)]
    time.sleep(0.01)
    t0 = time.perf_counter(); short = d.submit([seg(99)], [1], 3); short.result(timeout=5); dt_short = time.perf_counter() - t0
    assert not all(f.done() for f in long_f) and dt_short < 0.05
    [f.result(timeout=10) for f in long_f]
    d.close()
    assert not pool.decoder.on


def test_continuous_replica_picks_gemm_tiles_by_what_else_is_running():
    """A prefill on an idle replica asks for small GEMM tiles where the big ones under-fill the chip; beside running rows it leaves the idle
    CUs to the decode loop (gemm_small_eff 75 / 0); an explicit user setting switches the rule off."""
    pool = StubPool(n_rows=4, prefill_batch=4, prefill_delay=0.001, step_delay=0.002)
    d = Dispatcher([pool.decoder], slots=[pool.prefills], continuous=True)
    d.submit([seg(1)], [1], 8).result(timeout=5)
    assert pool.prefills[0].options == [("gemm_small_eff", 75)]
    long_f = d.submit([seg(2)], [1], 150)
    time.sleep(0.02)
    d.submit([seg(3)], [1], 4).result(timeout=5)
    assert pool.prefills[0].options[1:] == [("gemm_small_eff", 75), ("gemm_small_eff", 0)]
    long_f.result(timeout=10)
    d.close()
    pool = StubPool(n_rows=4, prefill_batch=4)
    d = Dispatcher([pool.decoder], slots=[pool.prefills], continuous=True, adaptive_tiles=False)
    d.submit([seg(1)], [1], 8).result(timeout=5)
    assert pool.prefills[0].options == []
    d.close()


def test_continuous_replica_two_decoders_share_the_prefill_slot():
    pool = StubPool(n_rows=3, prefill_batch=4, prefill_delay=0.001)
    dec2 = StubPool.Decoder(pool, 3)
    d = Dispatcher([pool.decoder], slots=[[dec2] + pool.prefills], continuous=True, decoders=2)
    budgets = [30 + (i * 7) % 40 for i in range(20)]
    futs = [d.submit([seg(i)], [1, 2, 3], mn) for i, mn in enumerate(budgets)]
    for i, (f, mn) in enumerate(zip(futs, budgets)):
        assert f.result(timeout=10).tolist() == want_tokens(i, 3, mn), i
    assert pool.decoder.max_occupied <= 3 and dec2.max_occupied <= 3 and dec2.max_occupied > 0       # both pools were used, neither overfilled
    assert d.replicas[0].load() == 0
    d.close()
    assert not pool.decoder.on and not dec2.on


def test_continuous_replica_errors_cancel_and_close():
    pool = StubPool(n_rows=2, prefill_batch=4, n_prefill=2, prefill_delay=0.005)
    d = Dispatcher([pool.decoder], slots=[pool.prefills], continuous=True)
    good = [d.submit([seg(i)], [1, 2], 20) for i in range(6)]
    bad = d.submit([seg(1, n=13)], [1, 2], 20)                          # the stub raises for this request only
    too_big = d.submit([seg(1)] * 5, [1], 20)                           # more windows than a prefill batch
    with pytest.raises(ValueError):
        bad.result(timeout=10)
    with pytest.raises(ValueError):
        too_big.result(timeout=10)
    assert [f.result(timeout=10).tolist() for f in good] == [want_tokens(i, 2, 20) for i in range(6)]
    assert pool.decoder.max_occupied <= 2
    assert d.replicas[0].load() == 0
    d.close()
    with pytest.raises(RuntimeError):
        d.submit([seg(0)], [1], 5)

    class Broken(StubPool.Decoder):
        def service_step(self, n_chunks=1, rows=0):
            raise RuntimeError("HIP error: device lost")
    pool2 = StubPool(n_rows=2)
    pool2.decoder = Broken(pool2, 2)
    d2 = Dispatcher([pool2.decoder], slots=[pool2.prefills], continuous=True)
    f = d2.submit([seg(0)], [1], 20)
    with pytest.raises(RuntimeError):
        f.result(timeout=10)
    time.sleep(0.05)
    with pytest.raises(RuntimeError):
        d2.submit([seg(0)], [1], 20)                                      # a failed engine refuses new work instead of hanging it
    d2.close()


def test_continuous_replica_close_while_a_prefill_is_in_flight():
    """close() 50 ms into a 200 ms prefill used to end the decode loop first; the prefill thread then span in _hand() for ever,
    close() blocked for its whole join timeout and the futures stayed pending.  Now a request whose rows a prefill has reserved completes,
    what is still queued fails with 'closed', and every thread is gone when close() returns."""
    pool = StubPool(n_rows=4, prefill_batch=2, prefill_delay=0.2)
    d = Dispatcher([pool.decoder], slots=[pool.prefills], continuous=True)
    futs = [d.submit([seg(i)], [1, 2], 6) for i in range(4)]             # two go into the first prefill batch, two stay queued behind it
    time.sleep(0.05)
    t0 = time.perf_counter()
    d.close()
    assert time.perf_counter() - t0 < 5.0
    assert all(f.done() for f in futs)
    done, failed = [], []
    for i, f in enumerate(futs):
        try:
            assert f.result(timeout=0).tolist() == want_tokens(i, 2, 6)
            done.append(i)
        except RuntimeError as ex:
            assert "closed" in str(ex)
            failed.append(i)
    assert done == [0, 1] and failed == [2, 3]
    assert not any(t.is_alive() for t in d.replicas[0].threads)
    assert not pool.decoder.on and all(r is None for r in pool.decoder.rows)


def test_continuous_replica_decoder_failure_while_a_prefill_is_in_flight():
    """a decode thread that fails while a prefill is still running leaves a hand-over nobody will take: the prefill side
    fails that batch itself.  With two decoders the sibling loop stops too instead of finishing futures that were already failed."""
    class Broken(StubPool.Decoder):
        def service_step(self, n_chunks=1, rows=0):
            raise RuntimeError("HIP error: device lost")
    pool = StubPool(n_rows=2, prefill_batch=1, prefill_delay=0.15)
    pool.decoder = Broken(pool, 2)
    dec2 = StubPool.Decoder(pool, 2)
    d = Dispatcher([pool.decoder], slots=[[dec2] + pool.prefills], continuous=True, decoders=2)
    fs = [d.submit([seg(i)], [1], 20) for i in range(4)]                 # one request per prefill batch: the second is in flight when the first one's decoder fails
    for f in fs:
        with pytest.raises(RuntimeError):
            f.result(timeout=10)
    t0 = time.perf_counter()
    d.close()
    assert time.perf_counter() - t0 < 5.0
    assert not any(t.is_alive() for t in d.replicas[0].threads)


def test_bulk_pipeline_pairs_batches_and_checks_every_row():
    """sonicscribe_amd/pipeline.py (bench.py's headline driver) over stub handles: every batch goes through prefill -> splice -> continuous
    decode -> fetch, rows are checked, blocks are reused, and an empty decoder leaves the next batch to one that runs part-filled."""
    from sonicscribe_amd.pipeline import ContinuousPipeline
    pool = StubPool(n_rows=4, prefill_batch=2, prefill_delay=0.001, step_delay=0.0005)
    dec2 = StubPool.Decoder(pool, 4)
    pre = pool.prefills[0]
    pipe = ContinuousPipeline([pool.decoder, dec2], [pre], block=2)
    assert pipe.batches_in_flight == 5 and pool.decoder.on and dec2.on
    n = {"prefills": 0}

    def prefill(p):
        n["prefills"] += 1
        p.stage_pcm([seg(7), seg(9)], [0, 1, 2])
        p.prefill([[1, 2, 3], [1, 2, 3]], [40, 24])

    want = [want_tokens(7, 3, 40), want_tokens(9, 3, 24)]
    res = pipe.run(9, prefill, lambda i, ids: ids.tolist() == want[i])
    assert res["batches"] == 9 and res["wrong_rows"] == 0 and n["prefills"] == 9
    assert pool.decoder.max_occupied <= 4 and dec2.max_occupied <= 4
    assert all(r is None for r in pool.decoder.rows) and all(r is None for r in dec2.rows)          # every row handed back
    assert max(pool.decoder.max_occupied, dec2.max_occupied) == 4                                   # two batches shared one loop
    bad = pipe.run(2, prefill, lambda i, ids: i != 1)                                               # a failing check is counted, not raised
    assert bad["batches"] == 2 and bad["wrong_rows"] == 2
    pipe.close()
    assert not pool.decoder.on and not dec2.on
    with pytest.raises(ValueError):
        ContinuousPipeline([], [pre])


class StubNativePipeline:
    """Stands in for pipeline.NativePipeline (the library's sonic_pipeline_*): tickets complete after `delay`, a batch that holds a 13-sample window
    fails as a whole with the engine's message, rows 'decode' like StubEngine's."""

    def __init__(self, decoders, prefills, block):
        self.decoders, self.prefills, self.block = list(decoders), list(prefills), block
        self.batches_in_flight = len(self.decoders) * 2 + len(self.prefills)
        self.submitted, self.closed, self.delay = [], False, 0.002
        self.results, self.lock = {}, threading.Lock()

    def submit(self, prompts, max_new, segments=None, req_win=None):
        if self.closed:
            raise RuntimeError("pipeline is closed")
        assert len(prompts) <= self.block and len(segments) <= self.block and req_win[0] == 0 and req_win[-1] == len(segments)
        with self.lock:
            t = len(self.submitted) + 1
            self.submitted.append((len(segments), list(max_new)))
        rows, err = [], None
        for r in range(len(prompts)):
            w = segments[req_win[r]:req_win[r + 1]]
            if any(len(x) == 13 for x in w):
                err = RuntimeError("Audio features and audio tokens do not match")
            rows.append(np.asarray([int(sum(int(x.sum()) for x in w)) % 1000, len(prompts[r]), max_new[r]], np.int32))
        self.results[t] = (time.perf_counter() + self.delay, rows, err)
        return t

    def wait(self, ticket):
        due, rows, err = self.results.pop(ticket)
        time.sleep(max(0.0, due - time.perf_counter()))
        if err is not None:
            raise err
        return rows

    def close(self):
        self.closed = True


def test_bulk_replica_groups_requests_into_native_pipeline_batches():
    """Dispatcher(bulk=True) -> dispatch._BulkReplica: requests become batches of up to `block` windows (oldest first, one step class per batch), go to
    the (stub) native pipeline, and every future gets its own row back; a request that fails its batch fails alone; close() drains."""
    eng, slots = StubEngine(max_batch=64), [StubEngine(64) for _ in range(3)]
    d = Dispatcher([eng], slots=[slots], bulk=True, decoders=3, pipeline_factory=StubNativePipeline)
    rep = d.replicas[0]
    pipe = rep.pipe
    assert len(pipe.decoders) == 3 and len(pipe.prefills) == 1 and pipe.block == 32
    fs = [d.submit([seg(i)], [1, 2, 3], 150) for i in range(80)] + [d.submit([seg(200 + i)], [1], 15) for i in range(5)]
    for i, f in enumerate(fs[:80]):
        assert f.result(timeout=10).tolist() == [(i * 16) % 1000, 3, 150]
    for i, f in enumerate(fs[80:]):
        assert f.result(timeout=10).tolist() == [((200 + i) * 16) % 1000, 1, 15]
    sizes = [n for n, _ in pipe.submitted]
    assert sum(sizes) == 85 and max(sizes) <= 32 and len(sizes) <= 6                       # full blocks while the queue is deep
    assert all(len(set(mn)) == 1 for _, mn in pipe.submitted)                              # a 15-token request never rides a 150-token batch
    # one bad request (the engine's validation error fails the whole ticket): its neighbours are resubmitted one by one and succeed
    n0 = len(pipe.submitted)
    good = [d.submit([seg(3)], [1], 150), d.submit([seg(4)], [1], 150)]
    bad = d.submit([seg(5, n=13)], [1], 150)
    assert [f.result(timeout=10).tolist() for f in good] == [[48, 1, 150], [64, 1, 150]]
    with pytest.raises(RuntimeError, match="do not match"):
        bad.result(timeout=10)
    assert len(pipe.submitted) >= n0 + 1
    # a multi-window request keeps its windows together; one that cannot fit into a block is refused
    two = d.submit([seg(1), seg(2)], [1, 2], 150)
    assert two.result(timeout=10).tolist() == [48, 2, 150]
    with pytest.raises(ValueError, match="windows"):
        d.submit([seg(1)] * 33, [1], 150).result(timeout=10)
    with pytest.raises(TypeError):
        d.submit([object()], [1], 150)                                                     # device ring slices belong to the row-level dispatcher
    t0 = time.perf_counter()
    d.close()
    assert time.perf_counter() - t0 < 5.0 and pipe.closed and not any(t.is_alive() for t in rep.threads)
    with pytest.raises(RuntimeError):
        d.submit([seg(1)], [1], 150)
    with pytest.raises(ValueError):
        Dispatcher([StubEngine(64)], slots=[[StubEngine(64)]], bulk=True, decoders=3, pipeline_factory=StubNativePipeline)   # no prefill slot left


def test_bulk_replica_close_while_a_batch_is_inside_submit():
    """close() 50 ms after 32 submits while pipe.submit takes 0.2 s.  The batch is between `q` and `inflight` when stop is set: the
    complete loop must stay until it has been waited for; requests still queued fail at once; no future is left pending and no thread survives."""
    class SlowSubmit(StubNativePipeline):
        def submit(self, *a, **k):
            time.sleep(0.2)
            return super().submit(*a, **k)

    for trial in range(3):
        d = Dispatcher([StubEngine(64)], slots=[[StubEngine(64) for _ in range(3)]], bulk=True, decoders=3, pipeline_factory=SlowSubmit)
        rep = d.replicas[0]
        fs = [d.submit([seg(i)], [1], 150) for i in range(32)] + [d.submit([seg(100 + i)], [1], 15) for i in range(4)]
        time.sleep(0.05)
        d.close()
        assert not any(t.is_alive() for t in rep.threads) and rep.pipe.closed
        done = 0
        for i, f in enumerate(fs):
            assert f.done(), f"future {i} left pending by close() (trial {trial})"
            if f.exception() is None:
                assert f.result().tolist() == [((i if i < 32 else 68 + i) * 16) % 1000, 1, 150 if i < 32 else 15]
                done += 1
            else:
                assert "closed" in str(f.exception())
        assert done >= 32                                                  # the batch that was being submitted completed normally
