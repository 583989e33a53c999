"""Which threads of the process burn host CPU while batches run (per-thread user+system time from /proc/self/task)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except OSError:
            pass
    return out


dims = replace(spec.FULL, eos_ids=())
e = Engine(dims, 0, max_batch=32, max_ctx=512)
e.load_synthetic(20260128)
n = 20 * 16000
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11]
e.stage_pcm([synth.synth_pcm(i, n) for i in range(32)])
e.run_staged([prompt] * 32, [150] * 32)
for mode in ("sync rerun_staged", "async + wait"):
    a = threads(); t0 = time.perf_counter()
    for _ in range(8):
        if mode.startswith("sync"):
            e.rerun_staged()
        else:
            e.run_staged_async(); e.wait()
    dt = time.perf_counter() - t0; b = threads()
    print(f"{mode}: {dt:.2f} s wall; main tid {os.getpid()}")
    for tid, (comm, cpu) in sorted(b.items(), key=lambda kv: -(kv[1][1] - a.get(kv[0], ("", 0))[1]))[:6]:
        d = cpu - a.get(tid, ("", 0))[1]
        if d > 0.02:
            print(f"   tid {tid} {comm:<20s} {d:.2f} CPU-s ({d / dt * 100:.0f} %)")
e.close()
