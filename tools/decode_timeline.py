"""In-kernel timeline of one decoder layer's decode kernels at the bench configuration (B=32, full dims): where inside each launch the
time goes.  Timestamps are the device's 100 MHz wall clock (10 ns), taken by thread 0 of every block.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine

B = int(os.environ.get("TIMELINE_B", "32"))      # 33 .. 64: the fused kernels' second pass shows as the last point of gate/up
dims = replace(spec.FULL, eos_ids=())
n_samples = 20 * 16000
n_audio = spec.audio_token_count(spec.valid_frames(n_samples))
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
e = Engine(dims, 0, max_batch=B, max_ctx=512)
e.load_synthetic(20260128)
for kv in sys.argv[1:]:
    k, v = kv.split("="); e.set_option(k, int(v))
e.stage_pcm([synth.synth_pcm(j, n_samples) for j in range(B)])
e.run_staged([prompt] * B, [80] * B)
e.set_option("ktrace", 14)
e.rerun_staged()
kt = e.debug_ktrace()
names = ["qkv (skinny_xs)", "attention", "o_proj (skinny_o)", "gate/up (skinny_gu)", "down (skinny_xs)"]
points = {0: ["entry", "loads issued", "X in LDS", "last W multiplied", "synced", "slabs written"],
          1: ["entry", "prologue done (slab sum, RoPE, append)", "KV loop done", "merged in LDS", "O written"],
          2: ["entry", "loads issued", "all landed", "synced", "MFMA done", "end"],
          3: (["entry", "up-front loads issued", "passes 0 / 1 normalised, passes 2 / 3 + all W requested", "passes 2 / 3 normalised (wave 0)", "all W landed (wave 0)",
               "MFMA done (wave 0)", "end (64 rows written)"] if B > 32 and "gu64_two_pass=1" not in sys.argv[1:] else       # skinny_gu64_kernel (round 5)
              ["entry", "first loads issued", "norm pass done, all W issued", "X + tile 0 landed (wave 0)", "X fragments read", "MFMA done (wave 0)", "end (rows 0-31 written)"]
              + (["second pass done (rows 32-63)"] if B > 32 else [])),
          4: ["entry", "loads issued", "X in LDS", "last W multiplied", "synced", "slabs written"]}
t_ref = None
prev_end = None
for s in range(5):
    a = kt[s]
    used = a[:, 0] > 0
    a = a[used].astype(np.float64)
    if t_ref is None:
        t_ref = a[:, 0].min()
    npts = len(points[s])
    print(f"{names[s]}: {used.sum()} blocks; first entry at {(a[:, 0].min() - t_ref) / 100:.2f} us"
          + (f" ({(a[:, 0].min() - prev_end) / 100:.2f} us after the previous kernel's last block ended)" if prev_end is not None else ""))
    k0 = a[:, 0].min()
    for i in range(npts):
        c = (a[:, i] - k0) / 100.0
        print(f"    {points[s][i]:42s} min {c.min():6.2f}  median {np.median(c):6.2f}  max {c.max():6.2f} us")
    prev_end = a[:, npts - 1].max()
print(f"layer span (qkv first entry -> down last end): {(prev_end - t_ref) / 100:.2f} us (+ add/RMSNorm, not instrumented)")
e.close()
