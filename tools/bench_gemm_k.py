import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
eng = Engine(spec.TINY, 0, max_batch=2, max_ctx=128); eng.load_synthetic(1)
for epi, name in ((0, "bias"), (1, "gelu"), (2, "resid")):
    for N in (5120, 1280):
        pts = []
        for K in (256, 512, 1280, 2560, 5120):
            ms = min(eng.bench_gemm(48000, N, K, epi, 10) for _ in range(2))
            pts.append((K, ms * 1e3))
        (k0, t0), (k1, t1) = pts[0], pts[-1]
        slope = (t1 - t0) / (k1 - k0); icpt = t0 - slope * k0
        tiles = 188 * (N // 256); rounds = tiles / 256
        print(f"epi={name:5s} N={N}: " + " ".join(f"K={k}:{t:.0f}us" for k, t in pts) + f" | slope {slope*64:.2f} us/Ktile-round-all, intercept {icpt:.0f} us = {icpt/rounds:.1f} us per block-round; main-loop rate {2*48000*N*64/(slope*64)/1e6:.0f} TF/s")
