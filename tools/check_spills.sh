#!/bin/bash
# Lists every kernel of the library whose assembly uses scratch memory (register spills, or - worse - an accumulator array the compiler could not keep
# in registers): hipcc -S of each .hip source, `; ScratchSize: N` with N > 0.  The hot kernels (gemm256_kernel, gemm_kernel, flash_enc_kernel,
# skinny_* of the shipped shapes) must not appear; a header change can do it silently (round 5: the int8 fc1 kernel, +20 % encoder time).
cd "$(dirname "$0")/../sonicscribe_amd/csrc" || exit 1
for f in gemm256 gemm attn attn_enc elementwise quant logmel ingest; do
  extra=""; case $f in attn|attn_enc) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fvisibility=hidden $extra -S --cuda-device-only $f.hip -o /tmp/spill_$f.s 2>/dev/null || { echo "$f.hip: compile failed"; continue; }
  awk -v f=$f '/^_Z[A-Za-z0-9_]*:|^[a-z_0-9]*:/{name=$1} /; ScratchSize: [1-9]/{print f ".hip", name, $0}' /tmp/spill_$f.s
done
