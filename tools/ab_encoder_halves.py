"""Would the encoder gain from running a batch of 32 as two independent half-batch chains on two streams (the partial last tile round of
one chain's GEMM filled by the other's tiles)?  Measured with what exists: one engine at B = 32 against two engines at B = 16 running
concurrently, one generated token each (mel + encoder + prefill only).   python tools/ab_encoder_halves.py   (GPU box)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine

dims = replace(spec.FULL, eos_ids=())
n_samples = 20 * 16000
n_audio = spec.audio_token_count(spec.valid_frames(n_samples))
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]

def make(B, seed):
    e = Engine(dims, 0, max_batch=B, max_ctx=512)
    e.load_synthetic(20260128)
    e.stage_pcm([synth.synth_pcm(seed + j, n_samples) for j in range(B)])
    e.run_staged([prompt] * B, [1] * B)
    return e

def run(engines, reps):
    def work(e):
        for _ in range(reps):
            e.rerun_staged()
    th = [threading.Thread(target=work, args=(e,)) for e in engines]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) / reps

whole = make(32, 0)
halves = [make(16, 100), make(16, 200)]
for _ in range(2):
    a = run([whole], 6)
    b = run(halves, 6)
    tw = whole.timings(); th = [h.timings() for h in halves]
    print(f"one chain, B = 32: {a * 1e3:.1f} ms per 32 segments (encoder {tw['encoder_ms']:.1f} + prefill {tw['prefill_ms']:.1f});  "
          f"two chains, B = 16 each: {b * 1e3:.1f} ms per 32 segments (per chain: encoder {th[0]['encoder_ms']:.1f} / {th[1]['encoder_ms']:.1f}, "
          f"prefill {th[0]['prefill_ms']:.1f} / {th[1]['prefill_ms']:.1f})", flush=True)
