python -m pytest tests/test_gpu_parity.py -q -x -k "fused_decode_rows or batch_matches" 2>&1 | tail -5
python tools/batch_vs_solo_b64.py native 2>&1 | tail -7
for o in "" "--opt no_fused_gu64=1"; do python bench.py --mode native --batch 64 --steps 4 --warmup 2 --slots 1 --no-cpu-baseline --no-extras $o 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 b64 $o', round(d['value'],1), {k: round(v,1) for k,v in d['stages_ms_per_step'].items()})"; done
