"""In-kernel timeline of the 256x256 GEMM on the encoder's shapes (option gemm_trace of sonic_bench_gemm): where a tile's fixed cost goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
eng = Engine(spec.TINY, 0, max_batch=2, max_ctx=128); eng.load_synthetic(1)
eng.set_option("gemm_trace", 1)
for name, M, N, K, epi in [("qkv_vt", 48000, 3840, 1280, 4), ("fc1", 48000, 5120, 1280, 1), ("fc2", 48000, 1280, 5120, 2), ("o", 48000, 1280, 1280, 2), ("bias", 48000, 5120, 1280, 0)]:
    ms = eng.bench_gemm(M, N, K, epi, 5)
    print(f"{name}: {ms * 1e3:.1f} us", flush=True)
