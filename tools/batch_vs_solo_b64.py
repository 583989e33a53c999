"""Batch invariance at FULL dimensions up to 64 rows: rows 0-3 of batches of 16 / 32 / 33 / 40 / 64 against their solo runs, max |dlogit|
(0.0 = the same bits).  Round 3: 0.08 beyond 32 rows (the unfused decode path summed in another order); round 4: the fused kernels cover
33 .. 64 rows.   python tools/batch_vs_solo_b64.py [native|int8]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace

import numpy as np

from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine, MODE_INT8, MODE_NATIVE

mode = sys.argv[1] if len(sys.argv) > 1 else "native"
d = replace(spec.FULL, eos_ids=())
e = Engine(d, 0, MODE_INT8 if mode == "int8" else MODE_NATIVE, max_batch=64, max_ctx=512)
e.load_synthetic(20260128)
n = 5 * 16000
segs = [synth.synth_pcm(200 + i, n) for i in range(64)]
n_audio = spec.audio_token_count(spec.valid_frames(n))
prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
solo = [e.transcribe_batch([s], [prompt], [6], want_logits=True) for s in segs[:4]]
worst_all = 0.0
for B in (16, 32, 33, 40, 64):
    ids, lg = e.transcribe_batch(segs[:B], [prompt] * B, [6] * B, want_logits=True)
    worst = max(float(np.abs(lg[:, r] - solo[r][1][:, 0]).max()) for r in range(4))
    worst_all = max(worst_all, worst)
    print(f"{mode} B={B}: max|dlogit| of rows 0-3 vs their solo runs {worst:.4f}")
print(f"worst {worst_all:.4f}")
e.close()
