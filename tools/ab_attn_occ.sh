#!/bin/bash
# A/B (round 6): decode attention compiled for 128 VGPRs (a second 8-wave block fits on its CU beside it; 21 dwords spilled) against the shipped 151-VGPR form.
# Headline = bulk pipeline (three 64-row loops + prefill slot), single = one batch of 32 at a time, slots = three whole batches in flight.  Alternating runs, one box.
for rep in 1 2 3; do
for v in "" "--opt decode_attn_occ2=1"; do
  python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 $v 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', 'headline', round(d['value'],1), 'slots', round(d['batches_in_flight_slots']['value'],1), 'single', round(d['single_batch']['value'],1), 'decode step ms', round(d['roofline']['avg_launch_ms'],4))"
done
done
