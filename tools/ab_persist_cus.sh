#!/bin/bash
# Bulk pipeline with the encoder / prefill GEMMs as persistent launches on fewer than all CUs (SONIC_AB build): do the decode loops' kernels run
# faster beside GEMMs that leave some CUs alone?   Output: gpurun_out/persist_cus.txt
out=gpurun_out/persist_cus.txt; mkdir -p gpurun_out; : > $out
run() { echo "== $1" >> $out; SONIC_TOOL_OPTS=$2 python tools/ab_continuous_throughput.py 64 1 24 2 2 2>&1 | grep -E "timed|Error|error" >> $out; }
run "default (one block per tile)" ""
run "persistent, 256 CUs" gemm256_persist=1
run "persistent, 240 CUs" gemm256_persist=1,gemm256_persist_cus=240
run "persistent, 224 CUs" gemm256_persist=1,gemm256_persist_cus=224
run "persistent, 192 CUs" gemm256_persist=1,gemm256_persist_cus=192
cat $out
