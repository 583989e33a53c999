python -m pytest tests/test_gpu_parity.py -q -k "logmel" 2>&1 | tail -2
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --slots 1 --pipeline off 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mel', d['mel_frontend'], d['stages_ms_per_step'])"
