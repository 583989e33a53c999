#!/bin/bash
# Busy-host A/B: bench.py alone, then beside 64 spinning processes (the GPU box's cgroup gives this job 16 CPUs, so the
# spinners eat the whole quota and every thread of the job is frozen for most of each 100 ms period).  Spinners are killed by PID.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/busy_host_ab.txt
: > $OUT
summ() { tail -1 | python3 -c '
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "batches_in_flight_slots": d.get("batches_in_flight_slots", {}).get("value"), "single_batch": d["single_batch"]["value"], "decode_ms": d["stages_ms_per_step"]["decode_ms"], "batches_in_flight": d["config"]["batches_in_flight"]}))'; }
B="python3 bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-extras --slots ${SLOTS:-3} ${EXTRA:-}"
echo "box: $(nproc) CPUs visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" >> $OUT
echo "idle host:      $($B 2>/dev/null | summ)" >> $OUT
PIDS=""
for i in $(seq ${SPINNERS:-64}); do ( while :; do :; done ) & PIDS="$PIDS $!"; done
sleep 1
echo "${SPINNERS:-64} spinners:    $($B 2>/dev/null | summ)" >> $OUT
kill $PIDS 2>/dev/null
wait 2>/dev/null
echo "idle again:     $($B 2>/dev/null | summ)" >> $OUT
cat $OUT
