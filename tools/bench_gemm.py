import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
eng = Engine(spec.TINY, 0, max_batch=2, max_ctx=128)
eng.load_synthetic(1)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shapes = [("qkv_vt", 48000, 3840, 1280, 4), ("fc1", 48000, 5120, 1280, 1), ("fc2", 48000, 1280, 5120, 2), ("qkv", 48000, 3840, 1280, 0), ("o", 48000, 1280, 1280, 2),
          ("prefill_gu", 8448, 12288, 2048, 3), ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0)]
variants = [("k256s", 0, 1), ("k256", 0, 0), ("k128", 1, 0)]
for name, M, N, K, epi in shapes:
    if which != "all" and which != name: continue
    line = f"{name:10s} M={M} N={N} K={K}: "
    best = {}
    for rnd in range(2):
        for vn, force, stg in variants:
            eng.set_option("gemm_force128", force); eng.set_option("gemm256_stagger", stg)
            ms = eng.bench_gemm(M, N, K, epi, iters)
            best[vn] = min(best.get(vn, 1e9), ms)
    for vn, _, _ in variants:
        line += f"{vn} {best[vn]*1e3:8.1f} us {2.0*M*N*K/best[vn]/1e9:7.1f} TF/s   "
    eng.set_option("gemm_force128", 0); eng.set_option("gemm256_stagger", 1)
    print(line, flush=True)
