"""How long are the LLM.int8 outlier lists of a decode step (one row = one reference call) with the bench's synthetic weights?
   python tools/i8_outlier_counts.py [batch]      (GPU box)"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from dataclasses import replace
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine, MODE_INT8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = replace(spec.FULL, eos_ids=())
e = Engine(d, 0, max_batch=B, max_ctx=512, mode=MODE_INT8)
e.load_synthetic(20260128)
n = 20 * 16000
segs = [synth.synth_pcm(200 + i, n) for i in range(B)]
n_audio = spec.audio_token_count(spec.valid_frames(n))
prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
e.transcribe_batch(segs, [prompt] * B, [12] * B)
for name, w in (("shn", d.dec_d), ("satt", d.dec_heads * d.dec_head_dim), ("sact", d.dec_ff)):
    x = e.debug_read(name, (B, w))
    c = (np.abs(x) >= 6.0).sum(axis=1)
    print(f"{name}: outliers per row min {c.min()} median {int(np.median(c))} mean {c.mean():.1f} max {c.max()}; absmax {np.abs(x).max():.2f}")
e.close()
