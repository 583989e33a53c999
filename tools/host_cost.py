"""Host side of the decode loop: CPU time inside hipGraphLaunch per token step for several chunk sizes, on an idle host and next to N
spinning processes (the 'busy host' of DESIGN.md 4 note 3).  Prints what the box gives this process first (CPUs, affinity, cgroup quota)."""
import os
import subprocess
import sys
import time
from dataclasses import replace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine


def box():
    out = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out[p] = open(p).read().strip()
        except OSError:
            pass
    try:
        out["loadavg"] = open("/proc/loadavg").read().strip()
    except OSError:
        pass
    return out


def main():
    print("box:", box(), flush=True)
    dims = replace(spec.FULL, eos_ids=())
    B, n_samples, max_new = 32, 20 * 16000, 150
    eng = Engine(dims, 0, max_batch=B, max_ctx=512)
    eng.load_synthetic(20260128)
    prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n_samples)) + [7, 301, 302, 303, 9, 11]
    eng.stage_pcm([synth.synth_pcm(i, n_samples) for i in range(B)])
    eng.run_staged([prompt] * B, [max_new] * B)

    def measure(tag, chunks=(1, 4, 16, 64)):
        for c in chunks:
            eng.set_option("decode_chunk", c)
            eng.set_option("decode_lookahead", 1)
            eng.rerun_staged()
            for rep in range(3):
                t0 = time.perf_counter(); eng.rerun_staged(); wall = (time.perf_counter() - t0) * 1e3
                t = eng.timings()
                print(f"{tag:>14} chunk {c:>2} run {rep}: wall {wall:7.1f} ms  decode {t['decode_ms']:7.1f} ms  host: prefill-enqueue {t['host_prefill_enqueue_ms']:6.1f}  "
                      f"graph launches {t['host_decode_launches']:>3} in {t['host_decode_launch_ms']:7.1f} ms  check waits {t['host_decode_wait_ms']:7.1f} ms  "
                      f"lookahead {t['decode_lookahead']}", flush=True)
    measure("idle", chunks=(4, 16))
    for n in (32, 64, 128):
        procs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(n)]
        time.sleep(1.0)
        try:
            measure(f"{n} spinners", chunks=(4, 16))
        finally:
            for p in procs:
                p.kill()
            for p in procs:
                p.wait()
    eng.close()


if __name__ == "__main__":
    main()
