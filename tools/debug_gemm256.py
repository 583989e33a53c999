"""Diagnostic: gemm256 shapes of tests/test_gpu_parity.py::test_gemm256_path, repeated, with the error footprint printed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine


def bf(x):
    return synth.round_bf16(np.asarray(x, np.float32))


def run(eng, tag):
    for (M, N, K, epi) in [(1024, 512, 256, 0), (2048, 256, 1280, 2), (600, 1280, 320, 0), (8320, 2048, 256, 2), (2048, 256, 1280, 0)]:
        rng = np.random.default_rng(M + N)
        A = bf(rng.standard_normal((M, K)) * 0.5); W = bf(rng.standard_normal((N, K)) * 0.2); b = bf(rng.standard_normal(N) * 0.1)
        lin = bf((A.astype(np.float64) @ W.T.astype(np.float64) + b).astype(np.float32))
        R = bf(rng.standard_normal((M, N)))
        ref = bf(lin + R) if epi == 2 else lin
        for rep in range(3):
            got = eng.test_gemm(A, W, b, resid=R if epi == 2 else None, epi=epi)
            bad = np.abs(got - ref) > 0.1
            if bad.any():
                rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
                print(f"[{tag}] {M}x{N}x{K} epi{epi} rep{rep}: BAD {int(bad.sum())} elems, rows {rows.min()}..{rows.max()} ({len(rows)}), cols {cols.min()}..{cols.max()} ({len(cols)}), "
                      f"max err {np.abs(got - ref).max():.3f}; row tiles {sorted(set((rows // 256).tolist()))[:10]} col tiles {sorted(set((cols // 256).tolist()))[:10]}; "
                      f"got==lin(no resid)? {bool(np.all(np.abs(got[bad] - lin[bad]) < 0.1))} got==0? {bool(np.all(got[bad] == 0))}", flush=True)
            else:
                print(f"[{tag}] {M}x{N}x{K} epi{epi} rep{rep}: ok", flush=True)


e = Engine(spec.TINY, 0, max_batch=16, max_ctx=512); e.load_synthetic(1)
run(e, "fresh")
if len(sys.argv) > 1:
    from dataclasses import replace
    d = spec.FULL
    big = Engine(d, 0, max_batch=32, max_ctx=512); big.load_synthetic(2)
    segs = [synth.synth_pcm(i, 320000) for i in range(32)]
    n_audio = spec.audio_token_count(spec.valid_frames(320000))
    prompt = [1, 17, 23, 5] + [d.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
    big.transcribe_batch(segs, [prompt] * 32, [3] * 32, want_logits=True)
    big.close()
    run(e, "after-full-engine")
    e2 = Engine(spec.TINY, 0, max_batch=16, max_ctx=512); e2.load_synthetic(1)
    run(e2, "new-engine-after-full")
