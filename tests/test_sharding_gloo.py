"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous segment sharding, per-rank work,
result gather in global order, and the bench's max-over-ranks timing reduce.  CPU only -- the per-rank "engine" is a stub that
returns a function of the segment index, so ordering mistakes are visible."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sonicscribe_amd.sharder import gather_results, shard_range


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_items, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_items, rank, world)
    local = [[i * 10 + 1, i * 10 + 2] for i in range(lo, hi)]          # "token ids" of segment i
    allr = gather_results(local, rank, world)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)                 # bench.py: MAX over ranks of the elapsed time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.asarray(allr + [[t.item(), float(hi - lo)]], dtype=np.float64))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [64, 7])
def test_two_rank_shard_and_gather(tmp_path, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    want = np.asarray([[i * 10 + 1, i * 10 + 2] for i in range(n_items)], dtype=np.float64)
    counts = 0
    for r in range(world):
        assert np.array_equal(got[r][:-1], want)          # every rank sees all segments in global order
        assert got[r][-1][0] == 0.5 + (world - 1)         # max over ranks
        counts += got[r][-1][1]
    assert counts == n_items                               # shards cover every segment exactly once
