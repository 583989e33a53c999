"""Bulk pipeline on one weight copy: decoding handles run continuous greedy loops over their rows (include/sonic_hip.h sonic_service_*), prefill
slots run log-mel + encoder + prompt forward + first token for whole batches and splice their rows into whichever decoder has a free block.

Why it is faster than whole batches in flight (bench.py's `batches_in_flight` object, ASRModel(continuous=False, slots=3)): a decode step costs
1.35 ms for 32 rows and 1.83 ms for 64, so two batches sharing one 64-row loop pay 0.92 ms per 32 rows and step; two such loops on their own
streams fill each other's launch gaps; the MFMA-bound prefill work runs beside them.  Every segment still gets the whole path (the reference's
`ASRModel.transcribe()`, backend/asr.py:335-488, per segment) and - rows being independent in every decode kernel up to 64 rows - exactly the
tokens of a solo run.

Two drivers of the same schedule:
  * NativePipeline (round 5, the headline leg of bench.py): the hand-overs are native threads inside libsonic_hip.so (csrc/pipeline.cpp,
    sonic_pipeline_*): this module only submits batches and waits for tickets;
  * ContinuousPipeline (round 4): Python threads over GIL-releasing C-ABI calls, one per decoder (splice, step, fetch), one per prefill slot -
    kept for A/B (`bench.py --pipeline-host python`, `tools/ab_continuous_throughput.py`) and for the CPU tests, which drive it with stub engines.
Request-level scheduling for live traffic is dispatch._ContinuousReplica.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading
import time
from typing import Any, Callable, List, Optional, Sequence

import numpy as np


class ContinuousPipeline:
    def __init__(self, decoders: Sequence[Any], prefills: Sequence[Any], block: int = 32, pair: bool = True):
        """decoders / prefills: engine handles sharing one weight copy (an Engine and its slot()s), all with the same max_batch; a decoder holds
        max_batch // block batches at a time.  Decoders are put into continuous mode here and taken out by close()."""
        if not decoders or not prefills:
            raise ValueError("at least one decoder and one prefill slot")
        self.decoders, self.prefills, self.block, self.pair = list(decoders), list(prefills), int(block), bool(pair)
        self.blocks = [list(range(i, i + self.block)) for i in range(0, (self.decoders[0].max_batch // self.block) * self.block, self.block)]
        if not self.blocks:
            raise ValueError("max_batch is smaller than a block")
        for d in self.decoders:
            d.service_begin()
        self.batches_in_flight = len(self.decoders) * len(self.blocks) + len(self.prefills)

    def close(self):
        for d in self.decoders:
            d.service_end()

    def run(self, n_batches: int, prefill: Callable[[Any], None], check: Optional[Callable[[int, np.ndarray], bool]] = None) -> dict:
        """n_batches batches of `block` requests through the pipeline.  prefill(slot): run the slot's staged batch up to its first tokens
        (Engine.prefill); it may return a tag that names the batch.  check(i, ids) - or check(i, ids, tag) for a tagged batch - for row i of
        every finished batch (returns False for a wrong row).  Returns wall time and counts."""
        ready: "queue.Queue" = queue.Queue()
        lock = threading.Lock()
        state = {"todo": n_batches, "done": 0, "bad": 0, "steps": 0}
        errors: List[BaseException] = []

        def prefiller(p):
            try:
                while not errors:
                    with lock:
                        if state["todo"] <= 0:
                            return
                        state["todo"] -= 1
                    tag = prefill(p)
                    ev = threading.Event()
                    ready.put((p, ev, tag))
                    while not ev.wait(0.5):                      # a decoder has queued the splice: the slot may overwrite its rows
                        if errors:
                            return
            except BaseException as ex:
                errors.append(ex)

        half = {}                                                # decoder index -> it has both running and free blocks right now

        def decoder(d):
            k = self.decoders.index(d)
            free = list(range(len(self.blocks)))
            occupied = {}                                        # block -> chunk sequence number after which its flags are valid
            tags = {}                                            # block -> the tag its prefill returned
            try:
                while not errors:
                    with lock:
                        if state["done"] >= n_batches:
                            return
                        half[k] = bool(occupied) and bool(free)
                        # an EMPTY loop leaves the next batch to a loop that is running part-filled: two batches in one 64-row loop read the
                        # weights once (1.83 ms per step), two part-filled loops read them twice (2 x 1.35 ms)
                        defer = self.pair and not occupied and any(v for j, v in half.items() if j != k)
                    if defer:
                        time.sleep(0.002)
                        continue
                    while free:
                        try:
                            p, ev, tag = ready.get(block=not occupied, timeout=0.02)
                        except queue.Empty:
                            break
                        free.sort()
                        b = free.pop(0)                          # lowest free block: the loop steps only as many rows as are occupied
                        occupied[b] = d.splice_rows(p, list(range(self.block)), self.blocks[b])
                        tags[b] = tag
                        ev.set()
                    if not occupied:
                        continue
                    fin, nn, seq, _ = d.service_step(1, (max(occupied) + 1) * self.block)
                    steps = 1
                    done_blocks = [b for b, va in occupied.items() if seq > va and all(fin[r] for r in self.blocks[b])]
                    if done_blocks and len(done_blocks) < len(occupied):
                        # fetching a block's rows takes this thread about as long as a chunk takes the device: the other block's next chunk goes
                        # out first, or the stream runs dry under the fetches (and the engine answers by queueing ever deeper)
                        d.service_step(1, (max(occupied) + 1) * self.block)
                        steps += 1
                    for b, va in list(occupied.items()):
                        if b in done_blocks:
                            bad = 0
                            got = d.fetch_rows(self.blocks[b], [int(nn[r]) for r in self.blocks[b]])
                            for i, ids in enumerate(got):
                                if check is not None and not (check(i, ids) if tags[b] is None else check(i, ids, tags[b])):
                                    bad += 1
                            del occupied[b]; free.append(b)
                            with lock:
                                state["done"] += 1; state["bad"] += bad
                    with lock:
                        state["steps"] += steps
            except BaseException as ex:
                errors.append(ex)

        threads = [threading.Thread(target=prefiller, args=(p,), name="sonic-pipe-prefill") for p in self.prefills]
        threads += [threading.Thread(target=decoder, args=(d,), name="sonic-pipe-decode") for d in self.decoders]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        if errors:
            raise errors[0]
        return {"wall_s": dt, "batches": state["done"], "wrong_rows": state["bad"], "decode_chunks": state["steps"]}


class NativePipeline:
    """The bulk pipeline with its host loop inside the library (include/sonic_hip.h sonic_pipeline_*; csrc/pipeline.cpp): submit() queues a
    batch and returns a ticket, native threads stage / prefill / splice / step / fetch, wait() blocks on a condition variable.  Same handles,
    same schedule and - rows being independent in every decode kernel - the same tokens as ContinuousPipeline."""

    def __init__(self, decoders: Sequence[Any], prefills: Sequence[Any], block: int = 32):
        if not decoders or not prefills:
            raise ValueError("at least one decoder and one prefill slot")
        self.decoders, self.prefills, self.block = list(decoders), list(prefills), int(block)
        self.lib = self.decoders[0].lib
        self.rows = (self.decoders[0].max_batch // self.block) * self.block
        if self.rows < self.block:
            raise ValueError("max_batch is smaller than a block")
        dec = (C.c_void_p * len(self.decoders))(*[d.h for d in self.decoders])
        pre = (C.c_void_p * len(self.prefills))(*[p.h for p in self.prefills])
        h = C.c_void_p()
        rc = self.lib.sonic_pipeline_create(dec, len(self.decoders), pre, len(self.prefills), self.block, self.rows, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"sonic_pipeline_create failed with status {rc}: " + (self.lib.sonic_last_error(self.decoders[0].h) or b"").decode())
        self.h = h
        self.batches_in_flight = len(self.decoders) * (self.rows // self.block) + len(self.prefills)
        self._keep = {}                                     # ticket -> arrays the native side reads / writes until the ticket is collected

    def _err(self) -> str:
        return (self.lib.sonic_pipeline_last_error(self.h) or b"").decode()

    def submit(self, prompts: Sequence[Sequence[int]], max_new: Sequence[int], segments: Optional[Sequence[np.ndarray]] = None,
               req_win: Optional[Sequence[int]] = None) -> int:
        """One batch of len(prompts) <= block requests.  segments: int16 PCM windows (staged by the pipeline's prefill thread), or None = the
        batch is what every prefill handle has staged already.  Returns the ticket."""
        from .engine import Engine, _p
        ids, poffs = Engine._pack_prompts(prompts)
        mn = np.ascontiguousarray(max_new, dtype=np.int32)
        rw = np.ascontiguousarray(req_win, dtype=np.int32) if req_win is not None else None
        R = len(prompts)
        ld = int(mn.max())
        out = np.zeros((R, ld), np.int32)
        out_len = np.zeros(R, np.int32)
        pcm = offs = None
        W = 0
        if segments is not None:
            pcm, offs = self.decoders[0]._pack_pcm(segments)
            W = len(segments)
        t = C.c_int64(0)
        rc = self.lib.sonic_pipeline_submit(self.h, _p(pcm), _p(offs), W, _p(rw), R, _p(ids), _p(poffs), _p(mn), _p(out), ld, _p(out_len), C.byref(t))
        if rc != 0:
            raise RuntimeError(self._err() or f"sonic_pipeline_submit failed with status {rc}")
        self._keep[t.value] = (pcm, offs, out, out_len)
        return int(t.value)

    def wait(self, ticket: int) -> List[np.ndarray]:
        """Blocks until the batch is complete; returns its rows' token ids.  Raises what the batch failed with."""
        rc = self.lib.sonic_pipeline_wait(self.h, int(ticket))
        _, _, out, out_len = self._keep.pop(ticket)
        if rc != 0:
            raise RuntimeError(self._err() or f"batch failed with status {rc}")
        return [out[r, : out_len[r]].copy() for r in range(len(out_len))]

    def stats(self) -> dict:
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        self.lib.sonic_pipeline_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"batches": int(a.value), "decode_chunks": int(b.value), "batches_in_flight": int(c.value)}

    def run(self, n_batches: int, prompts: Sequence[Sequence[int]], max_new: Sequence[int], check: Optional[Callable[[int, np.ndarray], bool]] = None) -> dict:
        """n_batches copies of one batch that every prefill handle has staged (bench.py's timed loop): everything is submitted at once, the clock
        runs until the last ticket is complete; rows are checked after it has stopped."""
        c0 = self.stats()["decode_chunks"]
        t0 = time.perf_counter()
        tickets = [self.submit(prompts, max_new) for _ in range(n_batches)]
        rows = [self.wait(t) for t in tickets]
        dt = time.perf_counter() - t0
        bad = 0
        if check is not None:
            for got in rows:
                bad += sum(0 if check(i, ids) else 1 for i, ids in enumerate(got))
        return {"wall_s": dt, "batches": len(rows), "wrong_rows": bad, "decode_chunks": self.stats()["decode_chunks"] - c0}

    def close(self):
        if self.h:
            self.lib.sonic_pipeline_destroy(self.h)
            self.h = None
