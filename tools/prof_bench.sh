#!/bin/bash
# default-size bench under the kernel trace: bench line + per-kernel summary into gpurun_out/<dir>; extra bench flags after the dir
set -u
OUT=gpurun_out/${1:-prof}; shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-two-chains --no-extras --steps 3 --warmup 1 "$@" > $ROOT/$OUT/bench_under_rocprof.json 2> /tmp/prof_kt.err
cd $ROOT
python tools/prof_summary.py /tmp/prof_kt 40 > $OUT/kernel_summary.txt 2>&1
head -34 $OUT/kernel_summary.txt
python - <<PY
import json
d=json.loads(open("$OUT/bench_under_rocprof.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages_ms_per_step"])
PY
