"""A/B of the fc1 GEMM (M=48000, N=5120, K=1280, bias+GELU epilogue): LDS table against arithmetic GELU.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine

d = replace(spec.FULL, enc_layers=1, dec_layers=1, vocab=1024, audio_token_id=1000, eos_ids=(990, 991, 992))
e = Engine(d, 0, max_batch=2, max_ctx=320)
e.load_synthetic(1)
for rep in range(3):
    for lut in (1, 0):
        e.set_option("no_gelu_lut", 1 - lut)
        ms = e.bench_gemm(48000, 5120, 1280, 1, 20)
        print(f"fc1 GEMM gelu_lut={lut}: {ms*1e3:.1f} us  {2*48000*5120*1280/ms/1e9:.0f} TFLOP/s", flush=True)
    ms = e.bench_gemm(48000, 5120, 1280, 0, 20)
    print(f"same GEMM, bias only: {ms*1e3:.1f} us", flush=True)
e.close()
