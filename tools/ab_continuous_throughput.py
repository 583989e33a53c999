"""Throughput of the bulk pipeline (sonicscribe_amd/pipeline.py) on the bench workload, swept over its shape: `n_decoders` handles decode
continuously over `rows` rows each, `n_prefill` slots of the same weights run log-mel + encoder + prefill for batches of 32 x 20 s segments and
splice their rows in as blocks free up.
  python tools/ab_continuous_throughput.py [rows=64] [n_prefill=1] [batches=32] [chunk=2] [n_decoders=2]
Every segment gets the full path (encoder, prefill, 150 greedy tokens); tokens are compared with a plain batch run of the same segments."""
import os
import sys
from dataclasses import replace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first, so that a run under rocprofv3 uses torch's bundled HIP runtime: tools/rocprof_runtime_repro.py)

from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine
from sonicscribe_amd.pipeline import ContinuousPipeline

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_prefill = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_batches = int(sys.argv[3]) if len(sys.argv) > 3 else 32
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n_dec = int(sys.argv[5]) if len(sys.argv) > 5 else 2
B, n_samples, max_new = 32, 20 * 16000, 150
dims = replace(spec.FULL, eos_ids=())
dec = Engine(dims, 0, max_batch=rows, max_ctx=512)
dec.load_synthetic(20260128)
dec.set_option("decode_chunk", chunk)
for kv in filter(None, os.environ.get("SONIC_TOOL_OPTS", "").split(",")):      # engine knobs for A/B runs: SONIC_TOOL_OPTS=key=int,key=int
    dec.set_option(kv.split("=")[0], int(kv.split("=")[1]))
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11]
segs = [synth.synth_pcm(i, n_samples) for i in range(B)]
want, _ = dec.transcribe_batch(segs, [prompt] * B, [max_new] * B)          # plain batch run: the tokens every row must reproduce
pre = [dec.slot() for _ in range(n_prefill)]
decs = [dec] + [dec.slot() for _ in range(n_dec - 1)]
for p in pre:
    p.stage_pcm(segs)                                                       # PCM stays staged on the slot
pipe = ContinuousPipeline(decs, pre, block=B, pair=os.environ.get("SONIC_PIPE_PAIR", "1") != "0")
run = lambda n: pipe.run(n, lambda p: p.prefill([prompt] * B, [max_new] * B, wait=False), lambda i, ids: np.array_equal(ids, want[i]))
for tag, nb in (("warm-up", pipe.batches_in_flight), ("timed", n_batches)):
    r = run(nb)
    print(f"{tag}: {n_dec} decoder(s) x {rows} rows, prefill slots {n_prefill}, chunk {chunk} ({pipe.batches_in_flight} batches in flight): {nb} batches of {B} in "
          f"{r['wall_s'] * 1e3:.0f} ms = {nb * B / r['wall_s']:.1f} segments/s ({r['wall_s'] / nb * 1e3:.1f} ms per batch, {r['decode_chunks']} chunks queued); rows differing from the plain batch run: {r['wrong_rows']}", flush=True)
pipe.close()
dec.close()
