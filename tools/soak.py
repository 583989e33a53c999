"""Randomised concurrency soak of the facade (run on the GPU box): T threads fire transcribe() / submit() / stream requests of random
length (0.1 - 40 s: one or two 30 s windows), budget and hotwords at a multi-replica ASRModel; every transcript must equal the one the
same model gave for that request alone beforehand (a segment's result does not depend on what it is batched with, bit for bit).
  python tools/soak.py [seconds] [threads] [mode] [tiny|full] [batch|continuous]
(batch: two replicas x two batch slots each; continuous: row-level scheduling, dispatch._ContinuousReplica)"""
import sys, os, time, threading, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sonicscribe_amd import spec, synth, frontend
from sonicscribe_amd.asr import ASRModel

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mode = sys.argv[3] if len(sys.argv) > 3 else "native"
full = len(sys.argv) > 4 and sys.argv[4] == "full"
continuous = len(sys.argv) > 5 and sys.argv[5] == "continuous"
m = ASRModel.from_synthetic(spec.FULL if full else spec.TINY, device="cuda:0,0", mode=mode, max_batch=8, max_ctx=1024, slots=2, continuous=continuous)
rng = random.Random(1234)
HOT = [None, ["alpha"], ["Beta", "gamma delta"], ["x"] * 3]
cases = []
for i in range(16 if full else 48):
    n = int(16000 * rng.choice([0.1, 0.3, 1.28, 2.0, 5.0, 7.7, 12.0, 20.0, 31.0, 40.0]))
    raw = (synth.synth_pcm(1000 + i, n).astype(np.float64) * rng.uniform(0.05, 1.0)).round().astype(np.int16)
    cases.append({"raw": raw, "max_new": rng.choice([1, 3, 8, 15, 24, 40]), "hot": rng.choice(HOT)})
t0 = time.time()
for c in cases:      # reference results, one request at a time
    c["want"] = m.transcribe(frontend.pcm_bytes_to_float(c["raw"].tobytes()), 16000, max_new_tokens=c["max_new"], hotwords=c["hot"])
print(f"{len(cases)} reference transcripts in {time.time() - t0:.1f} s", flush=True)
def free_mb():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        return fr.value / 2 ** 20 if hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0 else float("nan")
    except Exception:
        return float("nan")
free0 = free_mb()
stop = time.time() + seconds
errors, done = [], [0] * T

def worker(k):
    r = random.Random(k)
    streams = {}
    try:
        while time.time() < stop:
            c = r.choice(cases)
            how = r.choice(["transcribe", "submit", "stream", "stream", "batch"])
            if how == "transcribe":
                got = m.transcribe(frontend.pcm_bytes_to_float(c["raw"].tobytes()), 16000, max_new_tokens=c["max_new"], hotwords=c["hot"])
            elif how == "submit":
                got = m.submit(frontend.pcm_bytes_to_float(c["raw"].tobytes()), 16000, c["max_new"], c["hot"], session=f"s{k}").result(timeout=120)
            elif how == "batch":
                c2 = r.choice(cases)
                if c2["max_new"] and c2["hot"] == c["hot"]:
                    a, b = m.transcribe_batch([frontend.pcm_bytes_to_float(x["raw"].tobytes()) for x in (c, c2)], 16000, [c["max_new"], c2["max_new"]], c["hot"])
                    if b != c2["want"]:
                        errors.append((k, "batch second", b, c2["want"]))
                    got = a
                else:
                    continue
            else:          # streaming session: chunks of 1024 samples into a 45 s device ring, decode the newest len(raw) samples
                st = streams.get(k)
                if st is None:
                    st = streams[k] = m.open_stream(f"stream-{k}", buffer_seconds=45.0)
                data = c["raw"].tobytes()
                ids = [st.add_audio_chunk(data[i:i + 2048]) for i in range(0, len(data), 2048)]
                got = st.submit_chunks(ids[0], ids[-1], c["max_new"], c["hot"]).result(timeout=120)
            if got != c["want"]:
                errors.append((k, how, got, c["want"]))
            done[k] += 1
    except BaseException as ex:
        errors.append((k, "exception", repr(ex)))
    finally:
        for st in streams.values():
            st.close()

ts = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
[t.start() for t in ts]; [t.join() for t in ts]
print(f"{sum(done)} requests from {T} threads in {seconds:.0f} s, batches per replica {[r.batches for r in m._dispatcher.replicas]}, errors: {len(errors)}; "
      f"device memory free before / after the load: {free0:.0f} / {free_mb():.0f} MiB")
for e in errors[:10]:
    print("  ", e)
m.close()
sys.exit(1 if errors else 0)
