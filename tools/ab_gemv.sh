#!/bin/bash
# A/B (round 6): the 1 - 4 row token step as the GEMV chain (csrc/gemv.hip, option decode_gemv / ASRModel(low_latency=True)) against the default MFMA chain.
# B = 1 call shape (ASRModel.transcribe of 5 s / 75 tokens and 20 s / 150 tokens, one at a time) and 16 streaming sessions; one MI355X, full dims, alternating runs, p50 in ms.
for rep in 1 2; do
for v in "" "--opt decode_gemv=1"; do
  python bench.py --streaming --sessions 16 --ingest ring --slots 2 --continuous --single --opt decode_chunk=2 $v 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'single_5s', round(d['single_5s']['latency_ms']['p50'],1), 'single_20s', round(d['single_20s']['latency_ms']['p50'],1), 'partial p50', round(d['partial_latency_ms']['p50'],1), 'final p50/p99', round(d['final_latency_ms']['p50'],1), round(d['final_latency_ms']['p99'],1))"
done
done
