"""Scratch diagnostics run on the GPU box (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine

def bf(x): return synth.round_bf16(np.asarray(x, np.float32))
eng = Engine(spec.TINY, 0, max_batch=8, max_ctx=512)
eng.load_synthetic(20260128)
rng = np.random.default_rng(0)
for (M, N, K) in [(64, 128, 768), (32, 128, 768), (64, 128, 512), (64, 128, 256), (16, 64, 384), (48, 64, 384), (64, 64, 128), (64, 2048, 6144), (32, 2048, 6144)]:
    X = bf(rng.standard_normal((M, K))); W = bf(rng.standard_normal((N, K)) * 0.1)
    got = eng.test_skinny(X, W)
    ref = (X.astype(np.float64) @ W.T.astype(np.float64)).astype(np.float32)
    bad = np.abs(got - ref) > 1e-3 + 1e-4 * np.abs(ref)
    print(f"skinny {M}x{N}x{K}: bad {bad.sum()} / {bad.size}  maxerr {np.abs(got-ref).max():.4g}")
    if bad.any():
        rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
        print("  bad rows:", rows[:70], "n", len(rows)); print("  bad cols:", cols[:40], "n", len(cols))
        m, n = np.argwhere(bad)[0]
        print("  first bad", m, n, got[m, n], ref[m, n])
