import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine, MODE_INT8
from oracle import oracle as orc
def f16(x): return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
e = Engine(spec.TINY, 0, MODE_INT8, max_batch=8, max_ctx=512); e.load_synthetic(20260128)
M, N, K, group_rows = 16, 64, 128, 16
rng = np.random.default_rng(M + N + K)
X = f16(rng.standard_normal((M, K)) * 1.2); W = f16(rng.standard_normal((N, K)) * 0.06); b = f16(rng.standard_normal(N) * 0.1)
X[0, 3] = 6.0; X[M // 2, K - 1] = -11.5; X[M - 1, 64] = 7.25; X[M // 2, 17] = 5.99
got = e.test_linear_int8(X, W, b, group_rows=group_rows)
cb, scb = orc.quantize_rows(W)
ref = orc.linear_int8(X, cb, scb, b)
bad = got != ref
print("mismatch", int(bad.sum()), "max", float(np.abs(got - ref).max()), "where", list(zip(*np.where(bad)))[:8])
print("f16(5.99) =", float(f16(5.99)), "X vals", X[M // 2, 17], X[0, 3])
for (m, n) in list(zip(*np.where(bad)))[:4]:
    print(m, n, got[m, n], ref[m, n])
d = spec.TINY
t0 = time.time(); om = orc.Model(d, synth.synth_state_dict(d, 20260128, 2), mode=orc.MODE_INT8); print("model", time.time() - t0, flush=True)
seg = synth.synth_pcm(11, 320000)
t0 = time.time(); feats, mask = orc.logmel(seg); print("logmel", time.time() - t0, flush=True)
prompt = [1, 17, 23, 5] + [d.audio_token_id] * spec.audio_token_count(spec.valid_frames(len(seg))) + [7, 301, 302, 303, 9, 11]
t0 = time.time(); r = om.transcribe(feats, int(mask.sum()), prompt, 1); print("transcribe 1 tok", time.time() - t0, flush=True)
t0 = time.time(); r = om.transcribe(feats, int(mask.sum()), prompt, 8); print("transcribe 8 tok", time.time() - t0, flush=True)
om2 = orc.Model(d, synth.synth_state_dict(d, 20260128, True), mode=orc.MODE_BF16)
t0 = time.time(); r = om2.transcribe(feats, int(mask.sum()), prompt, 8); print("bf16 transcribe 8 tok", time.time() - t0, flush=True)
t0 = time.time(); ids, lg = e.transcribe_batch([seg], [prompt], [8], want_logits=True); print("gpu eager", time.time() - t0, flush=True)
t0 = time.time(); ids, lg = e.transcribe_batch([seg], [prompt], [8]); print("gpu graph", time.time() - t0, flush=True)
