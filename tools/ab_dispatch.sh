#!/bin/bash
# A/B of the row-level dispatchers on one MI355X (full dims): the library's native threads (csrc/dispatch.cpp, default) against the Python class
# (dispatch._ContinuousReplica).  facade: ASRModel.submit x 640 segments of 20 s; streaming: 128 and 16 sessions, device rings, real-time schedule.
OUT=gpurun_out/${1:-ab_dispatch}; mkdir -p $OUT
for rep in 1 2; do
for v in native python; do
  F=""; [ $v = python ] && F="--python-dispatch"
  python bench.py --facade-only $F 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('facade', '$v', d['dispatcher'], round(d['value'],1), 'segments/s')" | tee -a $OUT/ab.txt
done
done
for n in 128 16; do
for v in native python; do
  F=""; [ $v = python ] && F="--python-dispatch"
  python bench.py --streaming --sessions $n --ingest ring --continuous --slots 2 $F 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streaming $n', '$v', 'partial', {k: round(v,1) for k,v in d['partial_latency_ms'].items()}, 'final', {k: round(v,1) for k,v in d['final_latency_ms'].items()})" | tee -a $OUT/ab.txt
done
done
