for cfg in "--slots 2" "--slots 3 --decoders 2" "--slots 2 --batch 64" "--slots 3 --decoders 2 --batch 64"; do
  python bench.py --streaming --sessions 128 --ingest ring --continuous $cfg 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streaming 128 [$cfg]', 'partial', {k: round(v,1) for k,v in d['partial_latency_ms'].items()}, 'final', {k: round(v,1) for k,v in d['final_latency_ms'].items()})"
done
