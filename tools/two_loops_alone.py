"""How long does a 64-row decode step take when two such loops run side by side and NOTHING else is on the GPU?  (The bulk pipeline overlaps
its loops with the prefill of the next batch; a 64-row step's kernels then average 2.3x their solo time.)  n_loops handles each decode 64 rows
(two prefilled batches of 32 x 20 s) for 150 tokens; wall time from the first step to the last finished row.
    python tools/two_loops_alone.py [n_loops=2] [rows=64]"""
import os
import sys
import threading
import time
from dataclasses import replace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine

n_loops = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # unused handles (two streams each) created before the other loops' handles: shifts their hardware queues
B, n_samples, max_new = 32, 20 * 16000, 150
dims = replace(spec.FULL, eos_ids=())
root = Engine(dims, 0, max_batch=64, max_ctx=512)
root.load_synthetic(20260128)
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11]
segs = [synth.synth_pcm(i, n_samples) for i in range(B)]
pre = root.slot()
skipped = [root.slot() for _ in range(n_skip)]
decs = [root] + [root.slot() for _ in range(n_loops - 1)]
pre.stage_pcm(segs)
for rep in range(2):
    for d in decs:
        d.service_begin()
    seqs = []
    for d in decs:
        s = 0
        for blk in range(rows // B):
            pre.prefill([prompt] * B, [max_new] * B)
            s = d.splice_rows(pre, list(range(B)), list(range(blk * B, blk * B + B)))
        seqs.append(s)
    for d in decs:
        d.synchronize()
    steps = [0] * len(decs)

    def loop(k):
        d = decs[k]
        while True:
            fin, nn, seq, _ = d.service_step(1, rows)
            steps[k] += 1
            if seq > seqs[k] and all(fin[r] for r in range(rows)):
                return
    th = [threading.Thread(target=loop, args=(k,)) for k in range(len(decs))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for d in decs:
        d.fetch_rows(list(range(rows)), [max_new] * rows)
        d.service_end()
    print(f"[{n_skip} unused handles, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}] {n_loops} loop(s) x {rows} rows, {max_new} tokens, nothing else on the GPU: {dt * 1e3:.1f} ms = {dt * 1e3 / 149:.3f} ms per step and loop; "
          f"{n_loops * rows / B} batches -> {dt * 1e3 / (n_loops * rows / B):.1f} ms of decode per batch of 32 (chunks queued {steps})", flush=True)
root.close()
