for b in 1 2 4 8 16; do for o in "" "--opt gemm_force128=1"; do python bench.py --batch $b --slots 1 --pipeline off --steps 3 --warmup 2 --no-cpu-baseline --no-extras $o 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms_per_step']; print('B=$b $o', 'encoder', round(s['encoder_ms'],2), 'prefill', round(s['prefill_ms'],2), 'decode', round(s['decode_ms'],1))"; done; done
