#!/bin/bash
# Round-4 A/B runs on the GPU box (gpurun): slots 1/2/3/4, decode_chunk 1/4/8/16, SONIC_NO_UC, and the busy-host A/B (64 spinning
# processes beside bench.py).  Every line lands in gpurun_out/ab_round4.log; nothing here is a headline.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOG=gpurun_out/ab_round4.log
: > $LOG
run() { echo "### $*" >> $LOG; "$@" 2>> gpurun_out/ab_round4.err | tail -1 | python3 -c '
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print("PARSE FAIL", l[:200]); continue
    print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","single_batch","stages_ms_per_step")}), "roofline.frac", d["roofline"]["frac"], "in_flight", d["roofline"].get("in_flight",{}).get("frac"))
' >> $LOG; }
B="python3 bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras"
for s in 1 2 3 4; do run $B --slots $s; done
for c in 1 2 8 16; do run $B --slots 2 --opt decode_chunk=$c; done
echo "### SONIC_NO_UC=1" >> $LOG
SONIC_NO_UC=1 run $B --slots 2
# busy host: 64 spinning processes (killed by PID afterwards)
PIDS=""
for i in $(seq 64); do ( while :; do :; done ) & PIDS="$PIDS $!"; done
sleep 1
echo "### busy host (64 spinners)" >> $LOG
run $B --slots 2
run $B --slots 1
run $B --slots 2 --opt decode_chunk=1
kill $PIDS 2>/dev/null
wait 2>/dev/null
cat $LOG
