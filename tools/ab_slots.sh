for s in 2 3 4 5; do python3 bench.py --steps 16 --warmup 5 --no-cpu-baseline --no-extras --slots $s 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('slots', d['config']['batches_in_flight'], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'single', round(d['single_batch']['value'],1))"; done
