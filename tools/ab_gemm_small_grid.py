"""Which GEMM tile for small batches?  A 256x256 tile owns a CU; one 5 s request gives the encoder's linears 30 .. 120 such tiles on 256 CUs.
`gemm_small_eff` (sonic_set_option) prices the 128x128 kernel at that percentage of the big kernel's per-CU rate and routes a GEMM there when its
tile rounds come out fewer (gemm.hip small_grid_prefers128).  This sweeps it: encoder / prefill ms per batch size, native and int8.
    python tools/ab_gemm_small_grid.py [native|int8] [seconds]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace

from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine, MODE_INT8, MODE_NATIVE

mode = sys.argv[1] if len(sys.argv) > 1 else "native"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
d = replace(spec.FULL, eos_ids=())
e = Engine(d, 0, MODE_INT8 if mode == "int8" else MODE_NATIVE, max_batch=32, max_ctx=1024)
e.load_synthetic(20260128)
n = int(secs * 16000)
segs = [synth.synth_pcm(200 + i, n) for i in range(32)]
n_audio = spec.audio_token_count(spec.valid_frames(n))
prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
print(f"{mode}, {secs:g} s segments, prompt {len(prompt)} tokens;  ms per batch: encoder / prefill")
print("eff   " + "".join(f"{'B=' + str(B):>16}" for B in (1, 2, 4, 8, 16, 32)))
ref = {}
for eff in (0, 50, 60, 75, 90, 100):
    e.set_option("gemm_small_eff", eff)
    row = []
    for B in (1, 2, 4, 8, 16, 32):
        best = None
        for rep in range(4):
            ids, _ = e.transcribe_batch(segs[:B], [prompt] * B, [3] * B)
            t = e.timings()
            if rep and (best is None or t["encoder_ms"] + t["prefill_ms"] < best[0] + best[1]):
                best = (t["encoder_ms"], t["prefill_ms"])
        key = tuple(tuple(int(x) for x in r) for r in ids)
        assert ref.setdefault(B, key) == key, "tokens moved with the tile choice"
        row.append(best)
    print(f"{eff:<6}" + "".join(f"{a:8.2f}/{b:7.2f}" for a, b in row), flush=True)
e.close()
