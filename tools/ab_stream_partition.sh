#!/bin/bash
# Bulk pipeline (2 decoders x 64 rows + 1 prefill slot; handles are created in the order decoder, prefill, decoder): does giving the decode
# streams priority over the prefill stream, or fencing the two kinds of work onto their own CUs, raise the whole-GPU rate?
# Needs a `make -C sonicscribe_amd/csrc SONIC_AB=1` build (the env knobs are compiled out of the product library).
# Output: gpurun_out/stream_partition.txt
out=gpurun_out/stream_partition.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $1" >> $out; shift; env "$@" python tools/ab_continuous_throughput.py 64 1 24 2 2 2>&1 | grep timed >> $out; }
run "default" SONIC_EXP_NONE=1
run "decoders high priority" SONIC_EXP_PRIO=-1,1,-1
run "prefill high priority" SONIC_EXP_PRIO=1,-1,1
run "prefill on CUs 0-191, decoders everywhere" SONIC_EXP_CUS=,0-191,
run "prefill on CUs 0-223, decoders everywhere" SONIC_EXP_CUS=,0-223,
run "prefill 0-191, decoders 192-255" SONIC_EXP_CUS=192-255,0-191,192-255
run "prefill 0-191, decoders 128-255" SONIC_EXP_CUS=128-255,0-191,128-255
cat $out
