"""The four encoder GEMM shapes at the bench's M = 48000 through sonic_bench_gemm, with and without the staggered schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
e = Engine(d, 0, max_batch=2, max_ctx=320)
e.load_synthetic(1)
shapes = [("qkv+V^T", 48000, 3840, 1280, 4), ("out_proj", 48000, 1280, 1280, 2), ("fc1+GELU", 48000, 5120, 1280, 1), ("fc2", 48000, 1280, 5120, 2)]
tot = {0: 0.0, 1: 0.0}
for name, M, N, K, epi in shapes:
    line = f"{name:9s} M={M} N={N} K={K}:"
    for stg in (1, 0, 1, 0):
        e.set_option("gemm256_stagger", stg)
        ms = e.bench_gemm(M, N, K, epi, 20)
        line += f"  stagger={stg}: {ms * 1e3:7.1f} us ({2.0 * M * N * K / ms / 1e9:5.0f} TF/s)"
        tot[stg] += ms / 2
    print(line, flush=True)
flops = sum(2.0 * M * N * K for _, M, N, K, _ in shapes)
for stg in (1, 0):
    print(f"all four, stagger={stg}: {tot[stg] * 1e3:.1f} us per layer = {flops / tot[stg] / 1e9:.0f} TF/s = {flops / tot[stg] / 1e9 / 2500:.3f} of peak")
e.close()
