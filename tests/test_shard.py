"""N>1 path on CPU: shard-by-frame bookkeeping and the control-plane collectives over gloo,
world_size 2 (the GPU runs use the same code with backend nccl == RCCL)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_range_partitions_every_frame_once():
    from grayskull_amd.shard import frame_range, owner_of
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 8, 4096, 4099):
            seen = []
            for r in range(world):
                lo, hi = frame_range(r, world, total)
                seen += list(range(lo, hi))
                for f in range(lo, hi):
                    assert owner_of(f, world, total) == r
            assert seen == list(range(total))
            sizes = [frame_range(r, world, total)[1] - frame_range(r, world, total)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from grayskull_amd.shard import Sharder, frame_range
    sh = Sharder(backend="gloo")
    total = 7
    lo, hi = frame_range(rank, world, total)
    local = torch.tensor([10 * f + 1 for f in range(lo, hi)], dtype=torch.int32)
    sh.barrier()
    mx = sh.max_over_ranks(1.0 + rank)
    sm = sh.sum_over_ranks(hi - lo)
    allv = sh.all_gather_frames(local, total)
    q.put((rank, mx, sm, allv.tolist()))
    sh.close()


def test_gloo_world2_gathers_per_frame_results_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, mx, sm, allv in res:
        assert mx == 2.0 and sm == 7.0
        assert allv == [10 * f + 1 for f in range(7)]
