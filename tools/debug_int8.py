import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine, MODE_INT8
from oracle import oracle as orc

def f16(x): return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
e = Engine(spec.TINY, 0, MODE_INT8, max_batch=8, max_ctx=512); e.load_synthetic(1)
rng = np.random.default_rng(0)
for (M, N, K) in [(16, 64, 128), (16, 64, 256), (64, 64, 256), (600, 256, 512)]:
    X = f16(rng.standard_normal((M, K)) * 1.2); W = f16(rng.standard_normal((N, K)) * 0.06)
    cb, scb = orc.quantize_rows(W)
    for tag, Xc, bias in [("plain", X, None), ("bias", X, f16(rng.standard_normal(N) * 0.1)), ("outl", None, None)]:
        if Xc is None:
            Xc = X.copy(); Xc[0, 3] = 6.0; Xc[M // 2, K - 1] = -11.5
        got = e.test_linear_int8(Xc, W, bias)
        ref = orc.linear_int8(Xc, cb, scb, bias)
        bad = got != ref
        print(M, N, K, tag, "mismatch", int(bad.sum()), "of", bad.size, "max", float(np.abs(got - ref).max()), "got[0,:4]", got[0, :4], "ref[0,:4]", ref[0, :4], flush=True)
        if bad.any() and tag == "plain":
            rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
            print("   rows", rows[:10], "cols", cols[:10], "ratio", (got[bad] / ref[bad])[:6])

# outlier path dissection
M, N, K = 64, 64, 256
rng = np.random.default_rng(3)
X = f16(rng.standard_normal((M, K)) * 1.2); W = f16(rng.standard_normal((N, K)) * 0.06)
X[0, 3] = 6.0; X[M // 2, K - 1] = -11.5; X[M - 1, 64] = 7.25
cb, scb = orc.quantize_rows(W)
got = e.test_linear_int8(X, W, None); ref = orc.linear_int8(X, cb, scb, None)
oc = (np.abs(X) >= 6).any(0)
print("outlier cols", np.where(oc)[0], "mismatches", int((got != ref).sum()))
wdq = f16((cb.astype(np.float32) * scb[:, None]) * np.float32(7.874015718698502e-3))
for (m, n) in list(zip(*np.where(got != ref)))[:6]:
    a2 = np.float32(0)
    for k in np.where(oc)[0]:
        a2 = np.float32(a2 + np.float32(X[m, k] * wdq[n, k]))
    print(m, n, "got", repr(float(got[m, n])), "ref", repr(float(ref[m, n])), "a2", repr(float(a2)), "x outl", X[m, oc], "wdq", wdq[n, oc])
