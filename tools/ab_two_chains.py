"""How much of the decode loop's loss is latency that a second, independent chain on the same GPU can fill?  N engines on device 0,
each with its own staged batch, pulling steps from one shared counter (so they drift into whatever phase offset load gives them);
optional start stagger.  Run on the GPU box:  python tools/ab_two_chains.py [B] [total_steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
total = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dims = replace(spec.FULL, eos_ids=())
n_samples = 20 * 16000
n_audio = spec.audio_token_count(spec.valid_frames(n_samples))
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * n_audio + [7, 301, 302, 303, 9, 11]
engines = []
for i in range(3):
    e = Engine(dims, 0, max_batch=B, max_ctx=512)
    e.load_synthetic(20260128)
    segs = [synth.synth_pcm(100 * i + j, n_samples) for j in range(B)]
    e.stage_pcm(segs)
    e.run_staged([prompt] * B, [150] * B)
    engines.append(e)

def run(n, stagger_ms):
    left = [total]; lock = threading.Lock()
    def work(i):
        time.sleep(i * stagger_ms * 1e-3)
        while True:
            with lock:
                if left[0] == 0:
                    return
                left[0] -= 1
            engines[i].rerun_staged()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    tm = [engines[i].timings() for i in range(n)]
    print(f"{n} chain(s) x B={B}, stagger {stagger_ms} ms, {total} steps: {B * total / dt:.1f} segments/s  ({dt / total * 1e3:.1f} ms per step; last step per engine: "
          + "; ".join(f"enc {t['encoder_ms']:.0f} pre {t['prefill_ms']:.0f} dec {t['decode_ms']:.0f}" for t in tm) + ")", flush=True)

run(1, 0)
run(2, 0)
run(2, 150)
run(3, 0)
run(3, 100)
run(2, 150)
