"""Deterministic search for a batch composition whose per-request result differs from the request's solo result (full dimensions).
  python tools/find_batch_dependence.py [mode] [trials] [max_batch]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth, frontend
from sonicscribe_amd.asr import ASRModel

mode = sys.argv[1] if len(sys.argv) > 1 else "int8"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 150
MB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = ASRModel.from_synthetic(spec.FULL, device="cuda:0", mode=mode, max_batch=MB, max_ctx=1024)
rng = random.Random(1234)
cases = []
for i in range(16):
    n = int(16000 * rng.choice([0.1, 0.3, 1.28, 2.0, 5.0, 7.7, 12.0, 20.0, 31.0, 40.0]))
    raw = (synth.synth_pcm(1000 + i, n).astype(np.float64) * rng.uniform(0.05, 1.0)).round().astype(np.int16)
    cases.append({"wav": frontend.pcm_bytes_to_float(raw.tobytes()), "n": n, "max_new": rng.choice([1, 3, 8, 15, 24, 40, 90, 150])})
for c in cases:
    c["want"] = m.transcribe(c["wav"], 16000, max_new_tokens=c["max_new"])
    c["wins"] = len(frontend.split_windows(c["n"], spec.FULL))
bad = 0
for t in range(trials):
    k = rng.randint(2, max(6, MB - 2))
    pick = []
    w = 0
    for _ in range(k):
        c = rng.choice(range(len(cases)))
        if w + cases[c]["wins"] <= MB:
            pick.append(c); w += cases[c]["wins"]
    if len(pick) < 2:
        continue
    got = m.transcribe_batch([cases[c]["wav"] for c in pick], 16000, [cases[c]["max_new"] for c in pick])
    diff = [i for i, c in enumerate(pick) if got[i] != cases[c]["want"]]
    if diff:
        bad += 1
        print(f"trial {t}: batch {[(c, cases[c]['n'] / 16000, cases[c]['max_new']) for c in pick]} -> requests {diff} differ; "
              f"e.g. got {got[diff[0]][:40]!r} want {cases[pick[diff[0]]]['want'][:40]!r}", flush=True)
        again = m.transcribe_batch([cases[c]["wav"] for c in pick], 16000, [cases[c]["max_new"] for c in pick])
        print("   rerun identical to first:", again == got, "; rerun correct:", all(again[i] == cases[c]["want"] for i, c in enumerate(pick)), flush=True)
        if bad >= 6:
            break
print(f"{mode}: {bad} differing batches in {t + 1} trials")
m.close()
sys.exit(1 if bad else 0)
