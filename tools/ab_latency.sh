for v in 0 1; do
  if [ $v = 1 ]; then export SONIC_SPIN_SYNC=1; else unset SONIC_SPIN_SYNC; fi
  python bench.py --streaming --sessions 16 --ingest ring --slots 2 --continuous --single --opt decode_chunk=2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spin=$v', 'single_5s', round(d['single_5s']['latency_ms']['p50'],1), 'single_20s', round(d['single_20s']['latency_ms']['p50'],1), 'partial', round(d['partial_latency_ms']['p50'],1), 'final', round(d['final_latency_ms']['p50'],1), round(d['final_latency_ms']['p99'],1))"
done
unset SONIC_SPIN_SYNC
python bench.py --streaming --sessions 16 --ingest ring --slots 1 --single 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch-mode blocking', 'single_5s', round(d['single_5s']['latency_ms']['p50'],1), 'single_20s', round(d['single_20s']['latency_ms']['p50'],1), 'partial', round(d['partial_latency_ms']['p50'],1), 'final', round(d['final_latency_ms']['p50'],1), round(d['final_latency_ms']['p99'],1))"
