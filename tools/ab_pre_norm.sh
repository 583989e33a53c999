#!/bin/bash
# A/B of the <= 2-row decode step (round 6): five launches per layer (the q|k|v projection consumes down_proj's slabs itself) against the six-launch chain
# (option no_pre_norm=1: standalone add+RMSNorm), B = 1 call shape and 16 streaming sessions, one MI355X, full dims, alternating.
for rep in 1 2; do
for v in "" "--opt no_pre_norm=1"; do
  python bench.py --streaming --sessions 16 --ingest ring --slots 2 --continuous --single --opt decode_chunk=2 $v 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'single_5s', round(d['single_5s']['latency_ms']['p50'],1), 'single_20s', round(d['single_20s']['latency_ms']['p50'],1), 'partial p50', round(d['partial_latency_ms']['p50'],1), 'final p50/p99', round(d['final_latency_ms']['p50'],1), round(d['final_latency_ms']['p99'],1))"
done
done
