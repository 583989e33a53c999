"""Bulk pipeline on one weight copy: decoding handles run continuous greedy loops over their rows (include/sonic_hip.h sonic_service_*), prefill
slots run log-mel + encoder + prompt forward + first token for whole batches and splice their rows into whichever decoder has a free block.

Why it is faster than whole batches in flight (bench.py's `batches_in_flight` object, ASRModel(continuous=False, slots=3)): a decode step costs
1.35 ms for 32 rows and 1.83 ms for 64, so two batches sharing one 64-row loop pay 0.92 ms per 32 rows and step; two such loops on their own
streams fill each other's launch gaps; the MFMA-bound prefill work runs beside them.  Every segment still gets the whole path (the reference's
`ASRModel.transcribe()`, backend/asr.py:335-488, per segment) and - rows being independent in every decode kernel up to 64 rows - exactly the
tokens of a solo run.

The driver is host threads over GIL-releasing C-ABI calls: one per decoder (splice, step, fetch), one per prefill slot.  It is what bench.py times
for its headline and what `tools/ab_continuous_throughput.py` sweeps; request-level scheduling for live traffic is dispatch._ContinuousReplica.
"""
from __future__ import annotations

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
