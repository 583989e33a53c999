"""A/B of the int8 prefill operand (round 5): the fragment-tiled + k-major copies (shipped; the row-major int8 matrix is freed at load) against the
row-major matrix (SONIC_KEEP_ROWMAJOR=1 + option prefill_rowmajor) - same requests, logits compared bit for bit.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine, MODE_INT8
d = replace(spec.FULL, enc_layers=1, dec_layers=2, vocab=1024, audio_token_id=1000, eos_ids=())
R = 6
lens = [16000 * (1 + (i % 5)) + 37 * i for i in range(R)]
segs = [synth.synth_pcm(400 + i, n) for i, n in enumerate(lens)]
prompts = [[1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11][: 3 + i % 4] for i, n in enumerate(lens)]
out = {}
for mode in ("tiled", "rowmajor"):
    if mode == "rowmajor": os.environ["SONIC_KEEP_ROWMAJOR"] = "1"
    e = Engine(d, 0, MODE_INT8, max_batch=8, max_ctx=384)
    e.load_synthetic(11)
    if mode == "rowmajor": e.set_option("prefill_rowmajor", 1)
    ids, logits = e.transcribe_batch(segs, prompts, [3] * R, want_logits=True)
    out[mode] = (ids, logits, e.weight_bytes())
    e.close()
a, b = out["tiled"], out["rowmajor"]
print("weights", a[2] / 2**20, b[2] / 2**20)
print("ids equal", all(np.array_equal(x, y) for x, y in zip(a[0], b[0])))
dl = np.abs(a[1] - b[1])
print("max |dlogit|", dl.max(), "bit-identical", np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))
for s in range(a[1].shape[0]):
    print("step", s, "max diff per row", dl[s].max(axis=-1))
