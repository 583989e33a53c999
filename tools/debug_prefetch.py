import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine
d = replace(spec.FULL, enc_layers=1, dec_layers=2, vocab=1024, audio_token_id=1000, eos_ids=())
e = Engine(d, 0, max_batch=32, max_ctx=384)
e.load_synthetic(11)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
segs = [synth.synth_pcm(400 + i, 16000 * 2) for i in range(R)]
prompts = [[1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(len(s))) + [7, 9] for s in segs]
ids0, _ = e.transcribe_batch(segs, prompts, [6] * R)
for k in (4, 1, 2):
    e.set_option("no_graph", 1)
    e.set_option("decode_prefetch", k)
    print("running decode_prefetch =", k, flush=True)
    ids1, _ = e.transcribe_batch(segs, prompts, [6] * R)
    print("  ok, equal:", all(np.array_equal(a, b) for a, b in zip(ids0, ids1)), flush=True)
e.close()
