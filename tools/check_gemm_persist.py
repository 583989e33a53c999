"""gemm256p_kernel (persistent 256x256 GEMM, per-wave epilogue) against gemm256_kernel: bit-identical outputs for the bias / GELU / residual epilogues,
ragged M / N, several tiles per block (needs a SONIC_AB=1 build or the product build once the kernel ships)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine
e = Engine(spec.TINY, 0, max_batch=2, max_ctx=128); e.load_synthetic(1)
bf = lambda x: synth.round_bf16(np.asarray(x, np.float32))
rng = np.random.default_rng(0)
ok = True
for M, N, K, epi in [(1024, 512, 256, 0), (777, 384, 512, 1), (2048, 256, 1280, 2), (600, 1280, 320, 0), (70000, 512, 256, 1), (66000, 768, 256, 2), (5000, 1280, 1280, 2), (4096, 5120, 1280, 1)]:
    A = bf(rng.standard_normal((M, K)) * 0.5); W = bf(rng.standard_normal((N, K)) * 0.2); b = bf(rng.standard_normal(N) * 0.1)
    R = bf(rng.standard_normal((M, N))) if epi == 2 else None
    outs = []
    for p in (0, 1):
        e.set_option("gemm256_persist", p)
        outs.append(e.test_gemm(A, W, b, R, epi))
    same = np.array_equal(outs[0], outs[1])
    ok &= same
    print(f"M={M} N={N} K={K} epi={epi}: persistent == one-block-per-tile: {same}  (max |diff| {np.abs(outs[0] - outs[1]).max():.5f}, finite {np.isfinite(outs[1]).all()})", flush=True)
e.set_option("gemm256_persist", 0)
e.close()
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
