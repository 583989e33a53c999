"""rocprofv3 --kernel-trace segfaults inside the HIP runtime (ROCm 7.2's /opt/rocm/lib/libamdhip64.so.7, which libsonic_hip.so links) when a
full-size decode queues more than ~80 token steps of graph launches; the same run under torch's bundled runtime (import torch before the engine
loads: IMPORT_TORCH=1) or without the profiler is clean.  Round 4, MI355X box:
    full 32 32 2 150  -> rc 139          IMPORT_TORCH=1 full 32 32 2 150 -> ok          full 32 32 2 80 -> ok          tiny 64 32 2 150 -> ok
Tools that run under rocprofv3 therefore import torch first (bench.py always did).
    rocprofv3 --kernel-trace -d /tmp/pr -o kt --output-format csv -- python3 tools/rocprof_runtime_repro.py <full|tiny> <max_batch> <B> <decode_chunk|-1> <max_new>"""
import os, sys
if os.environ.get("IMPORT_TORCH"):
    import torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from sonicscribe_amd import spec, synth
from sonicscribe_amd.engine import Engine
dims_name, mb, B, chunk, max_new = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dims = replace(spec.FULL if dims_name == "full" else spec.TINY, eos_ids=())
e = Engine(dims, 0, max_batch=mb, max_ctx=512)
e.load_synthetic(20260128)
if chunk >= 0:
    e.set_option("decode_chunk", chunk)
n = 20 * 16000 if dims_name == "full" else 5 * 16000
prompt = [1, 17, 23, 5] + [dims.audio_token_id] * spec.audio_token_count(spec.valid_frames(n)) + [7, 301, 302, 303, 9, 11]
segs = [synth.synth_pcm(i, n) for i in range(B)]
ids, _ = e.transcribe_batch(segs, [prompt] * B, [max_new] * B)
print("ok", dims_name, mb, B, chunk, max_new, len(ids[0]), flush=True)
e.close()
