import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from dataclasses import replace
from sonicscribe_amd import spec, synth, frontend
from sonicscribe_amd.engine import Engine
d = replace(spec.FULL, eos_ids=())
e = Engine(d, 0, max_batch=64, max_ctx=512)
e.load_synthetic(20260128)
n = 5 * 16000
segs = [synth.synth_pcm(200 + i, n) for i in range(40)]
n_audio = spec.audio_token_count(spec.valid_frames(n))
prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
solo = [e.transcribe_batch([s], [prompt], [6], want_logits=True) for s in segs[:4]]
for B in (16, 32, 33, 40):
    ids, lg = e.transcribe_batch(segs[:B], [prompt] * B, [6] * B, want_logits=True)
    worst = max(float(np.abs(lg[:, r] - solo[r][1][:, 0]).max()) for r in range(4))
    print(f"B={B}: max|dlogit| of rows 0-3 vs their solo runs {worst:.4f}")
e.close()
