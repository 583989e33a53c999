"""The encoder / prefill GEMM shapes through sonic_bench_gemm with the persistent 256x256 kernel (gemm256p.hip) and with one block per tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
e = Engine(d, 0, max_batch=2, max_ctx=320)
e.load_synthetic(1)
shapes = [("qkv+V^T", 48000, 3840, 1280, 4), ("out_proj", 48000, 1280, 1280, 2), ("fc1+GELU", 48000, 5120, 1280, 1), ("fc2", 48000, 1280, 5120, 2),
          ("pf_qkv", 8192, 3072, 2048, 0), ("pf_gu", 8192, 12288, 2048, 3), ("pf_down", 8192, 2048, 6144, 2)]
tot = {0: 0.0, 1: 0.0}
for name, M, N, K, epi in shapes:
    line = f"{name:9s} M={M} N={N} K={K}:"
    for rep in range(2):
        for p in (1, 0):
            e.set_option("gemm256_persist", p)
            ms = e.bench_gemm(M, N, K, epi, 20)
            if rep == 1:
                line += f"  persist={p}: {ms * 1e3:7.1f} us ({2.0 * M * N * K / ms / 1e9:5.0f} TF/s)"
                if M == 48000:
                    tot[p] += ms
    print(line, flush=True)
flops = sum(2.0 * M * N * K for _, M, N, K, _ in shapes[:4])
for p in (1, 0):
    print(f"encoder layer, persist={p}: {tot[p] * 1e3:.1f} us = {flops / tot[p] / 1e9:.0f} TF/s = {flops / tot[p] / 1e9 / 2500:.3f} of peak")
e.close()
