"""Request dispatcher in front of the HIP engine replicas (SURVEY.md §8e "in-process dispatcher", §8f1 "cross-session coalescer").

The reference is ONE process with ONE model object (backend/main.py:1001 `workers: 1`, backend/models_manager.py:16-32) that is entered
  * synchronously on the asyncio event-loop thread, once per partial / final decode of every WebSocket session
    (backend/transcription_manager.py:43-65, called from backend/connection_manager.py:127-166, 169-245), and
  * from up to three executor threads in file mode (backend/main.py:429-430, 616-624),
so every decode of every session is serialised on one device.  Here one engine replica lives on every visible MI355X; each replica
has a worker thread that drains its queue into device batches:

  * no linger: a request that finds its replica idle starts at once; while a batch runs, arrivals queue up and form the next batch
    by themselves (batching comes from load, never from waiting);
  * buckets: a batch holds requests of one step class only (max_new_tokens <= 16 / <= 64 / <= 256 / more), so a 15-token partial
    (transcription_manager.py:24) never rides a 150-step final (:37); within a class the oldest request goes first;
  * placement: a request with a session key goes to replica hash(key) mod G (a session's partials and finals stay on one GPU), unless
    that replica's backlog exceeds the least-loaded one's by more than a batch, then it is rebalanced; keyless requests (file mode:
    independent segments) go to the least-loaded replica.  Segments are independent: no collective, no cross-replica state.

Slots: a replica may hold several engine handles that share ONE weight copy (Engine.slot(), sonic_slot_create) - one worker thread per
slot, all draining the replica's single queue.  While slot 0's batch is in its latency-bound decode loop, slot 1 takes the next batch
and its MFMA-bound encoder / prefill (and later its decode steps) fill the first one's bubbles; the reference does the same in spirit
in file mode (backend/main.py:429-445: three decodes in flight on one model object).  Batch formation is unchanged: whichever slot is
free takes the oldest request and everything of its step class that fits.

A window is either host PCM or a slice of a session's device ring (engine.RingSlice); ring requests are pinned to the
replica that owns the ring.  Engines are duck-typed (`max_batch`, `transcribe_batch(segs, prompts, max_new, req_win=...)`) so the dispatcher is testable on CPU
with stub engines; ctypes releases the GIL inside the real engine call, so G worker threads drive G GPUs concurrently.
"""
from __future__ import annotations

import threading
import time
import zlib
from concurrent.futures import Future
from typing import Any, List, Optional, Sequence

import numpy as np

STEP_CLASSES = (16, 64, 256)      # upper bounds of max_new_tokens per bucket; anything larger shares the last bucket


def step_class(max_new: int) -> int:
    for i, ub in enumerate(STEP_CLASSES):
        if max_new <= ub:
            return i
    return len(STEP_CLASSES)


class Request:
    __slots__ = ("windows", "prompt", "max_new", "future", "t_submit", "cls")

    def __init__(self, windows: Sequence[Any], prompt: Sequence[int], max_new: int):
        self.windows, self.prompt, self.max_new = list(windows), prompt, int(max_new)
        self.future: Future = Future()
        self.t_submit = time.perf_counter()
        self.cls = step_class(self.max_new)


class _Replica:
    def __init__(self, engine, index: int, slots: Sequence[Any] = ()):
        self.engine, self.index = engine, index
        self.engines = [engine] + list(slots)      # slot handles share engine's weights; each gets its own worker thread
        self.q: List[Request] = []
        self.cv = threading.Condition()
        self.stop = False
        self.busy = [0] * len(self.engines)        # windows of the batch each slot has on the device right now
        self.batches = 0
        self.threads = [threading.Thread(target=self._loop, args=(k,), name=f"sonic-replica-{index}.{k}", daemon=True) for k in range(len(self.engines))]
        for t in self.threads:
            t.start()

    @property
    def busy_windows(self) -> int:
        return sum(self.busy)

    def load(self) -> int:
        with self.cv:
            return self.busy_windows + sum(len(r.windows) for r in self.q)

    def put(self, req: Request):
        with self.cv:
            if self.stop:
                raise RuntimeError("ASR engine is closed")
            self.q.append(req)
            self.cv.notify()

    def _take(self, k: int = 0) -> List[Request]:
        with self.cv:
            while not self.q and not self.stop:
                self.cv.wait()
            # a caller may have cancelled a queued request (a disconnected session): it leaves the queue here and never reaches the device
            self.q = [r for r in self.q if not r.future.cancelled()]
            if not self.q:
                return []
            cap = self.engine.max_batch
            head = self.q[0]
            if len(head.windows) > cap:          # a single request larger than a device batch
                self.q.pop(0)
                if head.future.set_running_or_notify_cancel():
                    head.future.set_exception(ValueError(f"audio spans {len(head.windows)} windows, engine max_batch is {cap}"))
                return []
            batch, used, rest = [], 0, []
            for r in self.q:                     # oldest first; same step class as the head; whatever fits
                if r.cls == head.cls and used + len(r.windows) <= cap:
                    if r.future.set_running_or_notify_cancel():      # RUNNING: cancel() now returns False, set_result cannot raise
                        batch.append(r); used += len(r.windows)
                else:
                    rest.append(r)
            self.q = rest
            self.busy[k] = used
            if rest:
                self.cv.notify()                 # what did not fit (or is of another step class) is for the next free slot
            return batch

    @staticmethod
    def _finish(r: Request, result=None, error: Optional[BaseException] = None):
        if r.future.done():
            return
        if error is not None:
            r.future.set_exception(error)
        else:
            r.future.set_result(result)

    def _run(self, batch: List[Request], k: int = 0):
        engine = self.engines[k]
        segs, req_win = [], [0]
        for r in batch:
            segs.extend(r.windows)
            req_win.append(len(segs))
        try:
            ids, _ = engine.transcribe_batch(segs, [r.prompt for r in batch], [r.max_new for r in batch], req_win=req_win)
        except BaseException as ex:              # a per-request validation error must not poison its neighbours: retry one by one
            if len(batch) == 1:
                self._finish(batch[0], error=ex)
                return
            for r in batch:
                try:
                    one, _ = engine.transcribe_batch(r.windows, [r.prompt], [r.max_new], req_win=[0, len(r.windows)])
                except BaseException as ex2:
                    self._finish(r, error=ex2)
                else:
                    self._finish(r, one[0])
            return
        for r, i in zip(batch, ids):             # futures complete outside the engine's try block: a callback's error is not an engine error
            self._finish(r, i)

    def _loop(self, k: int = 0):
        while True:
            batch: List[Request] = []
            try:
                batch = self._take(k)
                if batch:
                    with self.cv:
                        self.batches += 1        # (counted before the futures complete: a waiter may read it right after its result)
                    self._run(batch, k)
            except BaseException as ex:          # the worker must not die silently: its queue would hang forever
                for r in batch:
                    try:
                        self._finish(r, error=ex)
                    except BaseException:
                        pass
            finally:
                with self.cv:
                    self.busy[k] = 0
            if not batch and self.stop and not self.q:
                return

    def close(self):
        with self.cv:
            self.stop = True
            self.cv.notify_all()
        for t in self.threads:
            t.join(timeout=30)
        for r in self.q:
            if not r.future.done():
                r.future.set_exception(RuntimeError("ASR engine is closed"))


class _Handover:
    __slots__ = ("engine", "reqs", "taken")

    def __init__(self, engine, reqs):
        self.engine, self.reqs, self.taken = engine, reqs, threading.Event()


class _ContinuousReplica:
    """Row-level scheduling on one weight copy (include/sonic_hip.h sonic_service_*): the replica's engine (and, with decoders > 1, that many
    of its slots) decodes FOREVER over its max_batch rows; the remaining slots only prefill.  A request goes: queue -> a prefill slot takes
    whatever is queued (no linger, any mix of step classes, as many requests as the emptiest decoder has free rows) -> log-mel, encoder,
    prompt forward, first token -> its row is spliced into a free row of that decoder between two chunks -> it leaves the moment it hits EOS / its budget.  Nobody waits for a running batch to
    end and no row idles until the slowest row of its batch is done - what the reference's per-connection `await transcribe()`
    (connection_manager.py:127-245) turns into when every session shares one device.  Tokens equal the solo run's bit for bit (decode rows are
    independent)."""

    def __init__(self, engine, index: int, slots: Sequence[Any], decoders: int = 1, adaptive_tiles: bool = True):
        decoders = max(1, int(decoders))
        # A prefill beside RUNNING rows keeps the big GEMM tiles even where they under-fill the chip (engine option gemm_small_eff = 0): the
        # idle CUs are where the decode loop's kernels run meanwhile.  Small tiles everywhere took a lone request's encoder 8.8 -> 7.2 ms but
        # the final p50 of 128 sessions 431 -> 490 ms (profiles/round4_streaming_ab.txt).  The tile choice never changes a result's bits.
        self.adaptive_tiles = bool(adaptive_tiles)
        if len(slots) < decoders:
            raise ValueError("continuous decoding needs at least one prefill slot per replica beside its decoding handles")
        self.engine, self.index = engine, index
        self.decoders = [engine] + list(slots[:decoders - 1])
        self.prefill_engines = list(slots[decoders - 1:])
        self.engines = [engine] + list(slots)
        self.q: List[Request] = []
        self.cv = threading.Condition()
        self.stop = False
        self.n_rows = engine.max_batch
        self.free = [self.n_rows] * len(self.decoders)   # per decoder: rows neither occupied nor reserved by a prefill in flight
        self.rows: List[List[Optional[Request]]] = [[None] * self.n_rows for _ in self.decoders]
        self.valid_after = [[0] * self.n_rows for _ in self.decoders]
        self.handovers: List[List[_Handover]] = [[] for _ in self.decoders]
        self.batches = 0                                 # prefill batches
        self.steps = 0                                   # decode chunks queued
        self.failed: Optional[BaseException] = None
        for d in self.decoders:
            d.service_begin()
        self.threads = [threading.Thread(target=self._decode_loop, args=(k,), name=f"sonic-decode-{index}.{k}", daemon=True) for k in range(len(self.decoders))]
        self.threads += [threading.Thread(target=self._prefill_loop, args=(k,), name=f"sonic-prefill-{index}.{k}", daemon=True) for k in range(len(self.prefill_engines))]
        for t in self.threads:
            t.start()

    @property
    def free_rows(self) -> int:
        return sum(self.free)

    def load(self) -> int:
        with self.cv:
            return (self.n_rows * len(self.decoders) - sum(self.free)) + sum(len(r.windows) for r in self.q)

    def put(self, req: Request):
        with self.cv:
            if self.stop:
                raise RuntimeError("ASR engine is closed")
            if self.failed is not None:
                raise RuntimeError(f"ASR engine failed: {self.failed}")
            self.q.append(req)
            self.cv.notify_all()

    _finish = staticmethod(_Replica._finish)

    # ---- prefill side
    def _take(self, cap: int):
        with self.cv:
            while not self.stop and (not self.q or max(self.free) <= 0):
                self.cv.wait()
            if self.stop:                                # closing: nothing new is started; close() fails what is still queued
                return [], 0
            self.q = [r for r in self.q if not r.future.cancelled()]
            k = max(range(len(self.free)), key=self.free.__getitem__)        # the emptiest decoder takes the whole prefill batch
            batch, used, rest = [], 0, []
            for r in self.q:                             # oldest first, whatever fits the slot's windows and the free rows; classes mix
                if len(r.windows) > cap:
                    if r.future.set_running_or_notify_cancel():
                        r.future.set_exception(ValueError(f"audio spans {len(r.windows)} windows, engine max_batch is {cap}"))
                elif len(batch) < self.free[k] and used + len(r.windows) <= cap and not rest:
                    if r.future.set_running_or_notify_cancel():
                        batch.append(r); used += len(r.windows)
                else:
                    rest.append(r)
            self.q = rest
            self.free[k] -= len(batch)                   # reserved until the rows are fetched (or the prefill fails)
            return batch, k

    def _prefill(self, eng, batch: List[Request]):
        segs, req_win = [], [0]
        for r in batch:
            segs.extend(r.windows)
            req_win.append(len(segs))
        if self.adaptive_tiles:
            with self.cv:
                busy = sum(self.n_rows - f for f in self.free) > len(batch)        # rows running or reserved besides this batch's own
            eng.set_option("gemm_small_eff", 0 if busy else 75)
        eng.stage_pcm(segs, req_win)
        # waited for: a splice queued behind a prefill that is still running would hold the decoder's whole stream (every running row) at the
        # event until the prefill is done - final p50 at 128 sessions 438 -> 657 ms when the hand-over came early.  (The bulk pipeline hands over
        # early, sonicscribe_amd/pipeline.py: its next batch is long done when a block frees up.)
        eng.prefill([r.prompt for r in batch], [r.max_new for r in batch], req_win)

    def _hand(self, eng, batch: List[Request], k: int):
        h = _Handover(eng, batch)
        with self.cv:
            self.handovers[k].append(h)
            self.cv.notify_all()
        while not h.taken.wait(0.5):                     # the slot's rows are the splice's source until the decode thread has queued it
            if self.failed is not None:                  # the decode side died after this batch was prefilled: its handler may have drained the
                with self.cv:                            # hand-over list before this one was appended - fail the batch here, exactly once
                    mine = h in self.handovers[k]
                    if mine:
                        self.handovers[k].remove(h)
                if mine:
                    for r in batch:
                        self._finish(r, error=self.failed)
                    self._release(k, len(batch))
                return

    def _release(self, k: int, n: int):
        with self.cv:
            self.free[k] += n
            self.cv.notify_all()

    def _prefill_loop(self, j: int):
        eng = self.prefill_engines[j]
        while True:
            batch, k = self._take(eng.max_batch)
            if not batch:
                if self.stop:
                    return
                continue
            with self.cv:
                self.batches += 1
            try:
                self._prefill(eng, batch)
            except BaseException as ex:                  # a per-request validation error must not poison its neighbours: one by one
                for r in batch:
                    try:
                        if len(batch) == 1:
                            raise ex
                        self._prefill(eng, [r])
                    except BaseException as ex2:
                        self._finish(r, error=ex2)
                        self._release(k, 1)
                    else:
                        self._hand(eng, [r], k)
                continue
            self._hand(eng, batch, k)

    # ---- decode side
    def _decode_loop(self, k: int):
        d, rows, valid_after = self.decoders[k], self.rows[k], self.valid_after[k]
        occupied = 0
        try:
            while True:
                with self.cv:
                    # idle = no row occupied and none reserved by a prefill in flight (free[k] counts both): a closing replica keeps its
                    # decode loop until every prefill that has taken rows has handed them over and they are fetched (close() used to
                    # leave a prefill thread spinning in _hand() with nobody left to take its rows)
                    while not self.handovers[k] and occupied == 0 and self.failed is None and not (self.stop and self.free[k] == self.n_rows):
                        self.cv.wait()
                    if self.failed is not None:
                        raise self.failed                # a sibling decoder failed: this loop's rows are failed by its own handler below
                    hs, self.handovers[k] = self.handovers[k], []
                    if self.stop and not hs and occupied == 0 and self.free[k] == self.n_rows:
                        return
                for h in hs:
                    free = [i for i, r in enumerate(rows) if r is None][:len(h.reqs)]
                    seq = d.splice_rows(h.engine, list(range(len(h.reqs))), free)
                    for i, r in zip(free, h.reqs):
                        rows[i], valid_after[i] = r, seq
                    occupied += len(h.reqs)
                    h.taken.set()
                if occupied == 0:
                    continue
                top = max(i for i, r in enumerate(rows) if r is not None) + 1          # rows are handed out lowest first: a light load stays in the first 16
                fin, nn, seq, _ = d.service_step(1, top)
                self.steps += 1
                done = [i for i, r in enumerate(rows) if r is not None and seq > valid_after[i] and fin[i]]
                if done:                                 # one call for all of them: one wait, one release launch (sonic_fetch_rows)
                    got = d.fetch_rows(done, [int(nn[i]) for i in done]) if len(done) > 1 else [d.fetch_row(done[0], int(nn[done[0]]))]
                    for i, ids in zip(done, got):
                        r, rows[i] = rows[i], None
                        occupied -= 1
                        self._finish(r, ids)
                    self._release(k, len(done))
        except BaseException as ex:                      # the engine failed: nothing queued or in flight can complete
            with self.cv:                                # every decode loop comes through here (siblings re-raise `failed`): each fails ITS rows
                if self.failed is None:
                    self.failed = ex
                ex = self.failed
                pending, self.q = self.q, []
                hs, self.handovers[k] = self.handovers[k], []
                mine, self.rows[k] = [x for x in rows if x is not None], [None] * self.n_rows
                self.cv.notify_all()
            for r in mine + [x for h in hs for x in h.reqs] + pending:
                try:
                    self._finish(r, error=ex)
                except BaseException:
                    pass
            for h in hs:
                h.taken.set()

    def close(self):
        with self.cv:
            self.stop = True
            self.cv.notify_all()
        for t in self.threads:
            t.join(timeout=30)
        for d in self.decoders:
            try:
                d.service_end()
            except BaseException:
                pass
        for r in self.q:
            if not r.future.done():
                r.future.set_exception(RuntimeError("ASR engine is closed"))


class _NativeContinuousReplica:
    """Row-level scheduling with the scheduler inside the LIBRARY (include/sonic_hip.h sonic_dispatch_*; csrc/dispatch.cpp): the same schedule as
    _ContinuousReplica - which stays as its executable description for stub engines (tests/test_dispatch.py) - run by native threads: prefill
    threads stage / prefill / hand over, decode threads splice / step / fetch.  Python keeps ONE thread per replica that collects completions
    (sonic_dispatch_next, blocking, GIL released) and resolves futures; put() is a single ctypes call.  No decode chunk passes the interpreter, so
    128 sessions' own Python work no longer delays the loops that serve them."""

    def __init__(self, engine, index: int, slots: Sequence[Any], decoders: int = 1, adaptive_tiles: bool = True):
        import ctypes as C
        decoders = max(1, int(decoders))
        if len(slots) < decoders:
            raise ValueError("continuous decoding needs at least one prefill slot per replica beside its decoding handles")
        self.engine, self.index = engine, index
        self.decoders = [engine] + list(slots[:decoders - 1])
        self.prefill_engines = list(slots[decoders - 1:])
        self.engines = [engine] + list(slots)
        self.n_rows = engine.max_batch
        self.lib = engine.lib
        dec = (C.c_void_p * len(self.decoders))(*[d.h for d in self.decoders])
        pre = (C.c_void_p * len(self.prefill_engines))(*[p.h for p in self.prefill_engines])
        h = C.c_void_p()
        rc = self.lib.sonic_dispatch_create(dec, len(self.decoders), pre, len(self.prefill_engines), int(bool(adaptive_tiles)), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"sonic_dispatch_create failed with status {rc}: " + (self.lib.sonic_last_error(engine.h) or b"").decode())
        self.h = h
        self.lock = threading.Lock()
        self.pending = {}                                # ticket -> Request
        self.stop = False
        self.out_cap = int(engine.max_ctx)
        self.thread = threading.Thread(target=self._complete_loop, name=f"sonic-dispatch-{index}.complete", daemon=True)
        self.thread.start()

    # diagnostics the Python class offers as attributes
    def _stats(self):
        import ctypes as C
        b, c, l, f = C.c_int64(0), C.c_int64(0), C.c_int32(0), C.c_int32(0)
        if self.h:
            self.lib.sonic_dispatch_stats(self.h, C.byref(b), C.byref(c), C.byref(l), C.byref(f))
        return int(b.value), int(c.value), int(l.value), int(f.value)

    @property
    def batches(self) -> int:
        return self._stats()[0]

    @property
    def steps(self) -> int:
        return self._stats()[1]

    @property
    def free_rows(self) -> int:
        return self._stats()[3]

    def load(self) -> int:
        return self._stats()[2]

    def put(self, req: Request):
        import ctypes as C
        from .engine import RingSlice, _p
        wins = req.windows
        W = len(wins)
        offs = np.zeros(W + 1, np.int64)
        host, rings, start, n, any_ring = [], [], np.zeros(W, np.int64), np.zeros(W, np.int32), False
        for i, w in enumerate(wins):
            if isinstance(w, RingSlice):
                if w.ring.engine.root is not self.engine.root:
                    raise ValueError("a ring slice can only be decoded by the engine that owns the ring (or by a slot of it)")
                rings.append(w.ring.h); start[i] = w.start; n[i] = w.n; any_ring = True
                offs[i + 1] = offs[i]
            else:
                a = np.ascontiguousarray(w, dtype=np.int16)
                host.append(a); rings.append(None)
                offs[i + 1] = offs[i] + len(a)
        pcm = np.concatenate(host) if host and offs[-1] > 0 else np.zeros(1, np.int16)
        ring_arr = (C.c_void_p * W)(*rings) if any_ring else None
        prompt = np.ascontiguousarray(req.prompt, dtype=np.int32)
        t = C.c_int64(0)
        with self.lock:
            if self.stop:
                raise RuntimeError("ASR engine is closed")
            rc = self.lib.sonic_dispatch_submit(self.h, _p(pcm), _p(offs), ring_arr, _p(start) if any_ring else None, _p(n) if any_ring else None, W,
                                                _p(prompt), len(prompt), int(req.max_new), C.byref(t))
            if rc != 0:
                raise RuntimeError("ASR engine failed" if rc not in (1,) else "ASR engine is closed or the request is malformed")
            self.pending[int(t.value)] = req
        ticket = int(t.value)
        req.future.add_done_callback(lambda f, ticket=ticket: f.cancelled() and self.h and self.lib.sonic_dispatch_cancel(self.h, ticket))

    def _complete_loop(self):
        import ctypes as C
        from .engine import SONIC_ERR_MISMATCH, SonicError
        ids = np.zeros(self.out_cap, np.int32)
        err = C.create_string_buffer(512)
        t, st, n = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        while True:
            rc = self.lib.sonic_dispatch_next(self.h, -1, C.byref(t), C.byref(st), ids.ctypes.data_as(C.c_void_p), self.out_cap, C.byref(n), err, 512)
            if rc != 0 or t.value == 0:
                return                                   # closed and drained
            with self.lock:
                req = self.pending.pop(int(t.value), None)
            if req is None or req.future.done():
                continue
            try:
                if st.value == 0:
                    req.future.set_result(ids[:n.value].copy())
                else:
                    msg = err.value.decode(errors="replace")
                    if "audio spans" in msg or st.value == SONIC_ERR_MISMATCH:
                        req.future.set_exception(ValueError(msg))
                    elif "closed" in msg:
                        req.future.set_exception(RuntimeError(msg))
                    else:
                        req.future.set_exception(SonicError(msg))
            except BaseException:                        # cancelled between the check and the set: the caller is gone
                pass

    def close(self):
        with self.lock:
            if self.stop:
                return
            self.stop = True
        self.lib.sonic_dispatch_close(self.h)            # queued requests fail, running ones complete
        self.thread.join()                               # ... and are delivered before the handle goes
        h, self.h = self.h, None
        self.lib.sonic_dispatch_destroy(h)
        for r in self.pending.values():
            if not r.future.done():
                r.future.set_exception(RuntimeError("ASR engine is closed"))
        self.pending.clear()


class _BulkReplica:
    """File mode (`backend/main.py:429-445`: whole segments submitted in bulk, three decodes kept in flight): requests are grouped, oldest first
    and one step class per batch, into batches of up to `block` windows and handed to the LIBRARY's pipeline (pipeline.NativePipeline ->
    csrc/pipeline.cpp: staging, prefill, splice, continuous decode loops and row fetches run in native threads).  Python keeps two threads per
    replica: one groups requests and submits batches, one waits for tickets and completes futures - nothing of the per-step schedule is here.
    A batch occupies a whole block of a decode loop until its last row has finished, so a trickle of single requests is better served by
    _ContinuousReplica (rows join and leave one by one); this class is for queues that are deep.  Measured (round 5, 1280 x 20 s segments through
    ASRModel.submit on one MI355X, idle host): 159 segments/s here against 169 for the row-level dispatcher - the Python threads of the latter are
    not what limits it; this form is the one whose per-step work does not depend on the interpreter at all."""

    def __init__(self, engine, index: int, slots: Sequence[Any], decoders: int = 3, block: int = 32, linger_s: float = 0.002, pipeline_factory=None):
        handles = [engine] + list(slots)
        if decoders < 1 or len(handles) < decoders + 1:
            raise ValueError("bulk mode needs `decoders` decoding handles and at least one prefill slot (slots >= decoders + 1)")
        if pipeline_factory is None:
            from .pipeline import NativePipeline as pipeline_factory
        self.engine, self.index, self.block, self.linger_s = engine, index, int(block), float(linger_s)
        self.pipe = pipeline_factory(handles[:decoders], handles[decoders:], self.block)
        self.max_in_flight = self.pipe.batches_in_flight + 2          # tickets submitted and not yet complete (bounds the host memory of a deep queue)
        self.q: List[Request] = []
        self.inflight: List[Any] = []                                 # (ticket, batch) in submission order
        self.submitting = 0                                           # batches taken from q and not yet in `inflight` (inside pipe.submit)
        self.cv = threading.Condition()
        self.stop = False
        self.batches = 0
        self.threads = [threading.Thread(target=self._submit_loop, name=f"sonic-bulk-{index}.submit", daemon=True),
                        threading.Thread(target=self._complete_loop, name=f"sonic-bulk-{index}.complete", daemon=True)]
        for t in self.threads:
            t.start()

    def load(self) -> int:
        with self.cv:
            return sum(len(r.windows) for r in self.q) + sum(sum(len(r.windows) for r in b) for _, b in self.inflight)

    def put(self, req: Request):
        if not all(isinstance(w, np.ndarray) for w in req.windows):
            raise TypeError("bulk mode takes host PCM windows (device ring slices are decoded by the row-level dispatcher: continuous=True, bulk=False)")
        with self.cv:
            if self.stop:
                raise RuntimeError("ASR engine is closed")
            self.q.append(req)
            self.cv.notify_all()

    _finish = staticmethod(_Replica._finish)

    def _take(self) -> List[Request]:
        """Oldest request first; everything of its step class that fits into one block.  Called with the lock held."""
        self.q = [r for r in self.q if not r.future.cancelled()]
        if not self.q:
            return []
        head = self.q[0]
        if len(head.windows) > self.block:
            self.q.pop(0)
            if head.future.set_running_or_notify_cancel():
                head.future.set_exception(ValueError(f"audio spans {len(head.windows)} windows, a pipeline block holds {self.block}"))
            return []
        batch, used, rest = [], 0, []
        for r in self.q:
            if r.cls == head.cls and used + len(r.windows) <= self.block:
                if r.future.set_running_or_notify_cancel():
                    batch.append(r); used += len(r.windows)
            else:
                rest.append(r)
        self.q = rest
        return batch

    def _submit(self, batch: List[Request]):
        segs, req_win = [], [0]
        for r in batch:
            segs.extend(r.windows)
            req_win.append(len(segs))
        return self.pipe.submit([r.prompt for r in batch], [r.max_new for r in batch], segments=segs, req_win=req_win)

    def _submit_loop(self):
        while True:
            with self.cv:
                while not self.stop and (not self.q or len(self.inflight) >= self.max_in_flight):
                    self.cv.wait()
                if self.stop:                                         # close() has failed what was still queued; nothing new is submitted
                    return
                if sum(len(r.windows) for r in self.q) < self.block:
                    self.cv.wait(self.linger_s)                       # a bulk submission arrives request by request: give the batch a moment to fill
                    if self.stop:
                        return
                batch = self._take()
                if batch:
                    self.submitting += 1                              # the complete loop must not leave while this batch is between q and inflight
            if not batch:
                continue
            try:
                ticket = self._submit(batch)
            except BaseException as ex:                               # refused at submission (closed pipeline, bad arguments): nobody else is affected
                with self.cv:
                    self.submitting -= 1
                    self.cv.notify_all()
                for r in batch:
                    self._finish(r, error=ex)
                continue
            with self.cv:
                self.inflight.append((ticket, batch))
                self.submitting -= 1
                self.batches += 1
                self.cv.notify_all()

    def _complete_loop(self):
        while True:
            with self.cv:
                while not self.inflight and not (self.stop and self.submitting == 0):
                    self.cv.wait()
                if not self.inflight:
                    return                                            # stopped, nothing in flight, nothing on its way into `inflight`
                ticket, batch = self.inflight[0]
            try:
                rows = self.pipe.wait(ticket)
            except BaseException as ex:
                rows, err = None, ex
            with self.cv:
                self.inflight.pop(0)
                self.cv.notify_all()
            if rows is not None:
                for r, ids in zip(batch, rows):
                    self._finish(r, ids)
            elif len(batch) == 1:
                self._finish(batch[0], error=err)
            else:
                # one request's validation error failed the batch (the engine's message names it): its neighbours go again, one batch each
                for r in batch:
                    try:
                        t = self._submit([r])
                        self._finish(r, self.pipe.wait(t)[0])
                    except BaseException as ex2:
                        self._finish(r, error=ex2)

    def close(self):
        """Requests still queued fail at once (as in _ContinuousReplica.close); batches already handed to the pipeline - including one that is
        inside pipe.submit right now - complete normally.  The native pipeline is destroyed only after both threads have left it."""
        with self.cv:
            self.stop = True
            queued, self.q = self.q, []
            self.cv.notify_all()
        for r in queued:
            if not r.future.done() and r.future.set_running_or_notify_cancel():
                r.future.set_exception(RuntimeError("ASR engine is closed"))
        for t in self.threads:
            t.join()                                                  # bounded by the batches in flight (at most max_in_flight), not by the queue
        self.pipe.close()


class Dispatcher:
    def __init__(self, engines: Sequence[Any], slots: Optional[Sequence[Sequence[Any]]] = None, continuous: bool = False, decoders: int = 1,
                 adaptive_tiles: bool = True, bulk: bool = False, pipeline_factory=None, native: Optional[bool] = None):
        """engines: one per replica (its own weights).  slots[i]: further engine handles that share replica i's weights (Engine.slot()).
        continuous: row-level scheduling (_ContinuousReplica: the engine - and decoders - 1 of its slots - decode forever, the other slots
        prefill) instead of batch by batch.  bulk: whole batches through the library's native pipeline (_BulkReplica)."""
        if not engines:
            raise ValueError("at least one engine")
        self.continuous = bool(continuous)
        if bulk:
            self.replicas = [_BulkReplica(e, i, slots[i] if slots else (), decoders, pipeline_factory=pipeline_factory) for i, e in enumerate(engines)]
        elif continuous:
            # native: the scheduler inside the library (csrc/dispatch.cpp) - the default for real engine handles; duck-typed stub engines (CPU tests)
            # and native=False keep the Python class, which is the same schedule statement by statement
            if native is None:
                native = all(hasattr(e, "h") and hasattr(getattr(e, "lib", None), "sonic_dispatch_create") for e in engines)
            cls = _NativeContinuousReplica if native else _ContinuousReplica
            self.replicas = [cls(e, i, slots[i] if slots else (), decoders, adaptive_tiles) for i, e in enumerate(engines)]
        else:
            self.replicas = [_Replica(e, i, slots[i] if slots else ()) for i, e in enumerate(engines)]

    def __len__(self):
        return len(self.replicas)

    def home(self, session: str) -> int:
        """Index of the replica a session key maps to (stable across processes, unlike hash())."""
        return zlib.crc32(str(session).encode()) % len(self.replicas)

    def pick(self, session: Optional[str]) -> _Replica:
        loads = [r.load() for r in self.replicas]
        least = min(range(len(loads)), key=loads.__getitem__)
        if session is None:
            return self.replicas[least]
        home = self.home(session)
        if loads[home] - loads[least] > self.replicas[home].engine.max_batch:
            return self.replicas[least]                                     # rebalance: the home replica is more than a batch behind
        return self.replicas[home]

    def submit(self, windows, prompt, max_new: int, session: Optional[str] = None, replica: Optional[int] = None) -> Future:
        """`replica` pins the request (windows that are slices of a device ring can only be decoded where the ring lives)."""
        req = Request(windows, prompt, max_new)
        (self.replicas[replica] if replica is not None else self.pick(session)).put(req)
        return req.future

    def close(self):
        for r in self.replicas:
            r.close()
