"""The four encoder GEMM shapes at the bench's M = 48000 through sonic_bench_gemm for several raster group heights (gemm256_gm)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
e = Engine(d, 0, max_batch=2, max_ctx=320)
e.load_synthetic(1)
shapes = [("qkv+V^T", 48000, 3840, 1280, 4), ("out_proj", 48000, 1280, 1280, 2), ("fc1+GELU", 48000, 5120, 1280, 1), ("fc2", 48000, 1280, 5120, 2)]
gms = [int(x) for x in (sys.argv[1:] or ["8", "4", "2", "16", "32", "188"])]
tot = {g: 0.0 for g in gms}
for name, M, N, K, epi in shapes:
    line = f"{name:9s} M={M} N={N} K={K}:"
    for rep in range(2):
        for gm in gms:
            e.set_option("gemm256_gm", gm)
            ms = e.bench_gemm(M, N, K, epi, 20)
            if rep == 1:
                line += f"  gm={gm}: {ms * 1e3:7.1f} us ({2.0 * M * N * K / ms / 1e9:5.0f} TF/s)"
                tot[gm] += ms
    print(line, flush=True)
flops = sum(2.0 * M * N * K for _, M, N, K, _ in shapes)
for gm in gms:
    print(f"all four, gm={gm}: {tot[gm] * 1e3:.1f} us per layer = {flops / tot[gm] / 1e9:.0f} TF/s = {flops / tot[gm] / 1e9 / 2500:.3f} of peak")
e.close()
