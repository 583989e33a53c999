#!/bin/bash
# streaming A/B runs (bench.py --streaming): `run <sessions> <flags...>` prints one summary line each
cd "$(dirname "$0")/.."
run() { python bench.py --streaming --sessions $1 --ingest ring "${@:2}" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', '| partial', {k: round(v,1) for k,v in d['partial_latency_ms'].items()}, 'final', {k: round(v,1) for k,v in d['final_latency_ms'].items()}, 'batches', d['device_batches_per_replica'])"; }
while read -r line; do [ -n "$line" ] && run $line; done
