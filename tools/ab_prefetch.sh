#!/bin/bash
# A/B (round 6): idle-CU weight prefetch in the 32-row decode step - option decode_prefetch: bit 0 attention launch -> o_proj weights, bit 1 -> first half of gate/up,
# bit 2 add+RMSNorm launch -> next q|k|v weights.  One batch in flight (the chain alone on the GPU): decode step in ms, single-batch segments/s.
for rep in 1 2; do
for k in 0 1 3 4 5 7; do
  python bench.py --no-cpu-baseline --no-extras --slots 1 --pipeline off --steps 6 --warmup 2 --opt decode_prefetch=$k 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('decode_prefetch=$k', 'decode step ms', round(d['roofline']['avg_launch_ms'],4), 'single', round(d['single_batch']['value'],2))"
done
done
