"""The real N-rank bench path on a one-GPU box (the 8-GPU scaling run itself is the driver's): bench.py under
`python -m torch.distributed.run --nproc-per-node 2`, gloo for the barrier / max-reduce (RCCL cannot put two ranks on one device),
both ranks on device 0 (--share-gpu).  Checks what the driver's SCALE run relies on: one JSON line from rank 0, n_gpus = 2, value =
the segments of BOTH ranks over the max-rank time, and every rank working on its own shard (sharder.shard_range).  This file sorts
first on purpose: the launcher process has not touched the GPU when it starts the child processes."""
import json
import os
import re
import subprocess
import sys

import pytest

from sonicscribe_amd.sharder import shard_range

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_one_gpu_through_torchrun():
    B, steps = 4, 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--dims", "tiny", "--batch", str(B), "--max-new", "12",
           "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--slots", "2", "--pipeline", "2x8+1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]                   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["config"]["share_gpu"] is True
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - 2 * B) < 1e-6 * 2 * B      # whole-job segments / max-over-ranks time
    assert out["config"]["batches_in_flight"] == 5 and out["config"]["weight_copies"] == 1           # 2 decoders x (8 rows / 4) + 1 prefill slot
    assert out["batches_in_flight_slots"]["slots"] == 2 and out["config"]["pipeline"]["rows_bit_identical_to_single_batch"] is True
    shards = {int(m.group(1)): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"\[bench\] rank (\d)/2 device 0 backend gloo segments \[(\d+), (\d+)\)", r.stderr)}
    assert shards == {0: shard_range(2 * B, 0, 2), 1: shard_range(2 * B, 1, 2)}, r.stderr[-1000:]
    assert shards[0][1] <= shards[1][0]                         # disjoint


def test_rccl_branch_at_world_size_one():
    """The `nccl` (= RCCL) branch of bench.py - init_process_group("nccl", device_id=...), the barrier and the device-tensor all_reduce of the
    elapsed time - executes once on the one-GPU box before the driver's 8-GPU SCALE run does: bench.py under torch.distributed.run with ONE
    rank (RCCL cannot put two ranks on one device).  The child is started before this process touches the GPU."""
    B, steps = 4, 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29733",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "1", "--dims", "tiny", "--batch", str(B), "--max-new", "12",
           "--dist-backend", "nccl", "--no-cpu-baseline", "--slots", "2", "--pipeline", "2x8+1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["config"]["dist_backend"] == "nccl"
    assert out["config"]["batches_in_flight"] == 5 and out["config"]["slots_bit_identical_to_single_batch"] is True
    assert out["config"]["pipeline"]["rows_bit_identical_to_single_batch"] is True
    assert abs(out["batches_in_flight_slots"]["value"] * out["batches_in_flight_slots"]["ms_per_step"] / 1e3 - B) < 1e-6 * B
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - B) < 1e-6 * B
    assert abs(out["single_batch"]["value"] * out["single_batch"]["ms_per_step"] / 1e3 - B) < 1e-6 * B
    assert "backend nccl" in r.stderr


def test_eight_ranks_share_one_gpu_and_leave_the_host_alone():
    """The driver's SCALE run puts 8 ranks on one node whose job has 16 CPUs of quota: every rank runs its pipeline threads
    (native since round 5: csrc/pipeline.cpp) plus the runtime's own.  Eight ranks of the real launch path on the one-GPU box (gloo, shared device,
    tiny dimensions): one JSON line, n_gpus = 8, disjoint shards, and a rank needs at most ~1.5 host CPUs in every leg - waits sleep (blocking events,
    condition variables), nothing polls - so 8 ranks stay within 12 CPUs."""
    B, steps, n = 4, 2, 8
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", "29735",
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", "1", "--dims", "tiny", "--batch", str(B), "--max-new", "12",
           "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--slots", "2", "--pipeline", "2x8+1"]
    # The host-CPU criterion is a load measurement over legs that last tens of milliseconds at these dimensions: on a cold box (eight interpreters paging
    # in torch at once) one leg has been seen above the bound once in five runs.  A second run on the then warm box must meet it; everything else is
    # asserted on every run.
    for attempt in range(2):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-1000:]
        out = json.loads(lines[0])
        if all(v <= 1.5 for v in out["config"]["host_cpus_busy"].values()):
            break
    assert out["n_gpus"] == n and out["config"]["share_gpu"] is True and out["config"]["pipeline"]["rows_bit_identical_to_single_batch"] is True
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - n * B) < 1e-6 * n * B
    assert "native threads" in out["config"]["pipeline"]["host"]
    shards = {int(m.group(1)): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"\[bench\] rank (\d)/8 device 0 backend gloo segments \[(\d+), (\d+)\)", r.stderr)}
    assert shards == {k: shard_range(n * B, k, n) for k in range(n)}, r.stderr[-1500:]
    busy = out["config"]["host_cpus_busy"]
    assert all(v <= 1.5 for v in busy.values()), busy          # x 8 ranks <= 12 CPUs of a 16-CPU quota
