"""Throughput of row-level scheduling on the bench workload: ONE engine decodes continuously over `rows` rows (sonic_service_*), `n_prefill`
slots of the same weights run log-mel + encoder + prefill for batches of 32 x 20 s segments and splice their rows in as blocks free up.
  python tools/ab_continuous_throughput.py [rows=64] [n_prefill=2] [batches=16] [chunk=2]
Every segment gets the full path (encoder, prefill, 150 greedy tokens); tokens are compared with a plain batch run of the same segments."""
import os
import queue
import sys
import threading
import time
from dataclasses import replace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_prefill = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_batches = int(sys.argv[3]) if len(sys.argv) > 3 else 16
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 2
B, n_samples, max_new = 32, 20 * 16000, 150
dims = replace(spec.FULL, eos_ids=())
dec = Engine(dims, 0, max_batch=rows, max_ctx=512)
dec.load_synthetic(20260128)
dec.set_option("decode_chunk", chunk)
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11]
segs = [synth.synth_pcm(i, n_samples) for i in range(B)]
want, _ = dec.transcribe_batch(segs, [prompt] * B, [max_new] * B)          # plain batch run: the tokens every row must reproduce
pre = [dec.slot() for _ in range(n_prefill)]
for p in pre:
    p.stage_pcm(segs)
    p.prefill([prompt] * B, [max_new] * B)                                  # warm-up (PCM stays staged)
dec.service_begin()
blocks = [list(range(i, i + B)) for i in range(0, rows, B)]                 # row blocks of 32
free_blocks = list(range(len(blocks)))
ready = queue.Queue()
todo = [n_batches]
lock = threading.Lock()


def prefiller(p):
    while True:
        with lock:
            if todo[0] <= 0:
                return
            todo[0] -= 1
        p.prefill([prompt] * B, [max_new] * B)
        ev = threading.Event()
        ready.put((p, ev))
        ev.wait()                                                           # the decode thread has queued the splice: the slot may go on


def run(n_batches_total):
    occupied = {}                                                           # block -> valid_after
    done, bad = 0, 0
    while done < n_batches_total:
        while free_blocks and not ready.empty():
            p, ev = ready.get()
            b = free_blocks.pop(0)
            occupied[b] = dec.splice_rows(p, list(range(B)), blocks[b])
            ev.set()
        if not occupied:
            p, ev = ready.get()                                             # nothing to decode yet: wait for the first prefill
            ready.put((p, ev))
            continue
        fin, nn, seq, _ = dec.service_step(1)
        for b, va in list(occupied.items()):
            if seq > va and all(fin[r] for r in blocks[b]):
                for i, r in enumerate(blocks[b]):
                    ids = dec.fetch_row(r, int(nn[r]))
                    bad += int(not np.array_equal(ids, want[i]))
                del occupied[b]; free_blocks.append(b); done += 1
    return bad


for tag, nb in (("warm-up", 2 * len(blocks)), ("timed", n_batches)):
    todo[0] = nb
    ts = [threading.Thread(target=prefiller, args=(p,)) for p in pre]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    bad = run(nb)
    dt = time.perf_counter() - t0
    [t.join() for t in ts]
    print(f"{tag}: rows {rows}, prefill slots {n_prefill}, chunk {chunk}: {nb} batches of {B} in {dt * 1e3:.0f} ms = {nb * B / dt:.1f} segments/s "
          f"({dt / nb * 1e3:.1f} ms per batch); rows differing from the plain batch run: {bad}", flush=True)
dec.service_end()
dec.close()
