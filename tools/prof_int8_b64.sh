#!/bin/bash
# int8 batch-64 bench under the kernel trace: bench line + per-kernel summary into gpurun_out/<dir>; extra bench flags after the dir
set -u
OUT=gpurun_out/${1:-i8prof}; shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_i8
rocprofv3 --kernel-trace -d /tmp/prof_i8 -o i8 --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-two-chains --no-extras --mode int8 --batch 64 --steps 2 --warmup 1 "$@" > $ROOT/$OUT/bench_int8_b64_under_rocprof.json 2> /tmp/prof_i8.err
cd $ROOT
python tools/prof_summary.py /tmp/prof_i8 30 > $OUT/int8_b64_kernel_summary.txt 2>&1
head -24 $OUT/int8_b64_kernel_summary.txt
python - <<PY
import json
d=json.loads(open("$OUT/bench_int8_b64_under_rocprof.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages_ms_per_step"])
PY
