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
