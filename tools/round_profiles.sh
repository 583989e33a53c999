#!/bin/bash
# Produces the measurement files of a round on the GPU box (run through gpurun from the repo root):
#   tools/round_profiles.sh <out dir under gpurun_out> [full]
# kernel trace (rocprofv3 --kernel-trace) of the bench with ONE batch in flight (--slots 1 --pipeline off: the leg the bench line's roofline is measured
# in, the decode kernels alone on the GPU) and of the default bench (all three legs: single batch, three whole batches in flight, the bulk pipeline), the FETCH_SIZE / WRITE_SIZE passes
# of a short bench (separate runs, counters only), the decode-step traffic derived from them, the int8 batch-64 kernel summary and (with `full`)
# the default bench line with every extra object.
set -u
OUT=gpurun_out/${1:-r3prof}
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_kt3 /tmp/prof_f /tmp/prof_w /tmp/prof_i8
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --slots 1 --pipeline off --steps 5 --warmup 1 > $ROOT/$OUT/bench_under_rocprof.json 2> /tmp/prof_kt.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt3 -o kt --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 3 > $ROOT/$OUT/bench_default_legs_under_rocprof.json 2> /tmp/prof_kt3.err
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --slots 1 --pipeline off --steps 1 --warmup 1 --max-new 150 > /dev/null 2> /tmp/prof_f.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --slots 1 --pipeline off --steps 1 --warmup 1 --max-new 150 > /dev/null 2> /tmp/prof_w.err
rocprofv3 --kernel-trace -d /tmp/prof_i8 -o i8 --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --slots 1 --pipeline off --mode int8 --batch 64 --steps 2 --warmup 1 > $ROOT/$OUT/bench_int8_b64_under_rocprof.json 2> /tmp/prof_i8.err
cd $ROOT
python tools/prof_summary.py /tmp/prof_kt 45 > $OUT/kernel_trace_summary.txt 2>&1
python tools/prof_summary.py /tmp/prof_kt3 45 > $OUT/kernel_trace_summary_default_legs.txt 2>&1
python tools/pmc_summary.py /tmp/prof_f > $OUT/pmc_fetch_size.txt 2>&1
python tools/pmc_summary.py /tmp/prof_w > $OUT/pmc_write_size.txt 2>&1
python tools/decode_traffic.py $OUT/pmc_fetch_size.txt $OUT/pmc_write_size.txt 28 150 > $OUT/pmc_decode.json 2>&1
python tools/prof_summary.py /tmp/prof_i8 45 > $OUT/int8_b64_kernel_summary.txt 2>&1
if [ "${2:-}" = "full" ]; then
  python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err      # the driver's command line
fi
head -30 $OUT/kernel_trace_summary.txt
cat $OUT/pmc_decode.json | head -8
