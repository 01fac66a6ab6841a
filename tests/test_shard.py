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


def _pipeline_worker(rank, world, port, q, emu_so):
    """one rank of the sharded config-2 / config-5 chains: its frames through the batch entry points (the
    kernel sources in their emulator build -- no GPU here), per-frame results gathered over gloo"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import grayskull_amd as G
    from grayskull_amd.shard import Sharder, frame_range
    from oracle.pyoracle import Oracle
    g = G.Grayskull(emu_so)
    sh = Sharder(backend="gloo")
    total, w, h = 5, 64, 24
    lo, hi = frame_range(rank, world, total)
    n = hi - lo
    src = np.stack([Oracle.synth(w, h, 1000 + f) for f in range(lo, hi)]) if n else np.zeros((0, h, w), np.uint8)
    dst = np.zeros_like(src)
    hist = np.zeros((max(n, 1), 256), np.uint32)
    thr = np.zeros(max(n, 1), np.uint8)
    if n:
        g.edge_pipeline_batch(dst, None, src, 2, hist, thr)        # fused blur -> sobel -> otsu -> threshold
        g.sync()
    sums = torch.tensor([int(Oracle.fnv1a(dst[i])) for i in range(n)], dtype=torch.int64)
    sh.barrier()
    all_thr = sh.all_gather_frames(torch.from_numpy(thr[:n].astype(np.int32)), total)
    all_sum = sh.all_gather_frames(sums, total)
    q.put((rank, all_thr.tolist(), all_sum.tolist()))
    sh.close()


def test_gloo_world2_sharded_pipeline_equals_oracle_on_every_frame():
    """BASELINE configs[1]/[4] sharding, world_size 2 on CPU: frames split by frame_range, each rank
    runs its share through gsh_edge_pipeline_batch, rank-ordered all-gather of the per-frame Otsu
    thresholds and output hashes; every rank sees all 5 frames' results == the oracle's"""
    import subprocess
    import numpy as np
    from oracle.pyoracle import Oracle
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "emu"])
    emu_so = os.path.join(ROOT, "tests", "emu", "libgs_kernel_emu.so")
    o = Oracle("port")
    exp_thr, exp_sum = [], []
    for f in range(5):
        e = o.sobel(o.blur(Oracle.synth(64, 24, 1000 + f), 2))
        t = o.otsu_threshold(e)
        exp_thr.append(int(t))
        exp_sum.append(int(Oracle.fnv1a(o.threshold(e, t))))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q, emu_so)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, thr, sums in res:
        assert thr == exp_thr, "rank %d thresholds" % rank
        assert sums == exp_sum, "rank %d output hashes" % rank


def _collectives_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from grayskull_amd.shard import Sharder, frame_range
    sh = Sharder(backend="gloo")
    blob = sh.broadcast_bytes(b"LBPC" + bytes(range(200)) if rank == 0 else b"", root=0)
    total, cap = 5, 6
    lo, hi = frame_range(rank, world, total)
    counts = torch.tensor([(3 * f) % (cap + 1) for f in range(lo, hi)], dtype=torch.int32)
    recs = torch.full((hi - lo, cap, 4), -1, dtype=torch.int32)        # rows past counts[f] hold junk (-1)
    for i, f in enumerate(range(lo, hi)):
        for k in range(int(counts[i])):
            recs[i, k] = torch.tensor([f, k, f * 100 + k, 7])
    call, rall = sh.gather_varlen(counts, recs, total)
    ok = sh.min_over_ranks(1.0 if rank == 0 else 0.0)
    q.put((rank, blob, call.tolist(), rall.tolist(), ok))
    sh.close()


def test_gloo_world2_cascade_broadcast_and_variable_length_gather():
    """SURVEY 8(e) collectives (1) and (3): every rank ends up with rank 0's blob and with every frame's
    result list, packed, in global frame order -- whatever the per-frame lengths are (incl. 0)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_collectives_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    exp_counts = [(3 * f) % 7 for f in range(5)]
    exp_recs = [[f, k, f * 100 + k, 7] for f in range(5) for k in range(exp_counts[f])]
    for rank, blob, call, rall, ok in res:
        assert blob == b"LBPC" + bytes(range(200))
        assert call == exp_counts and rall == exp_recs
        assert ok == 0.0


def _run_world(target, world, port, extra=(), timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=timeout) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res)


def test_gloo_world8_every_collective_with_ranks_that_own_no_frame():
    """the world size north_star is quoted on (8 GPUs), on CPU over gloo: the 7- and 5-frame jobs of the world-2 tests split
    over EIGHT ranks -- ranks 5..7 (resp. 7) own no frame and must still pass every collective: barrier, max / sum, the
    rank-ordered all-gather of per-frame results, the cascade broadcast and the packed variable-length gather"""
    res = _run_world(_worker, 8, 35500 + (os.getpid() % 2000))
    for rank, mx, sm, allv in res:
        assert mx == 8.0 and sm == 7.0
        assert allv == [10 * f + 1 for f in range(7)]
    res = _run_world(_collectives_worker, 8, 37500 + (os.getpid() % 2000))
    exp_counts = [(3 * f) % 7 for f in range(5)]
    exp_recs = [[f, k, f * 100 + k, 7] for f in range(5) for k in range(exp_counts[f])]
    for rank, blob, call, rall, ok in res:
        assert blob == b"LBPC" + bytes(range(200))
        assert call == exp_counts and rall == exp_recs
        assert ok == 0.0


def test_gloo_world8_sharded_pipeline_equals_oracle_on_every_frame():
    """configs[1] sharding at world 8 on CPU (kernel sources in their emulator build): 5 frames over 8 ranks, three of
    them idle; every rank sees all frames' Otsu thresholds and output hashes == the oracle's"""
    import subprocess
    from oracle.pyoracle import Oracle
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "emu"])
    emu_so = os.path.join(ROOT, "tests", "emu", "libgs_kernel_emu.so")
    o = Oracle("port")
    exp_thr, exp_sum = [], []
    for f in range(5):
        e = o.sobel(o.blur(Oracle.synth(64, 24, 1000 + f), 2))
        t = o.otsu_threshold(e)
        exp_thr.append(int(t))
        exp_sum.append(int(Oracle.fnv1a(o.threshold(e, t))))
    for rank, thr, sums in _run_world(_pipeline_worker, 8, 39500 + (os.getpid() % 2000), extra=(emu_so,)):
        assert thr == exp_thr, "rank %d thresholds" % rank
        assert sums == exp_sum, "rank %d output hashes" % rank


def test_forced_world1_uses_the_same_collectives(monkeypatch):
    """GS_BENCH_FORCE_DIST=1: a process group even at world 1 (gloo here, nccl == RCCL on the GPU box)"""
    from grayskull_amd.shard import Sharder
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("GS_BENCH_FORCE_DIST", "1")
    sh = Sharder(backend="gloo")
    try:
        assert sh.dist is not None and sh.world == 1 and sh.backend == "gloo" and sh.ranks_seen() == 1
        assert sh.broadcast_bytes(b"abc") == b"abc"
        c, r = sh.gather_varlen(torch.tensor([2, 0, 1]), torch.arange(3 * 4 * 2).reshape(3, 4, 2), 3)
        assert c.tolist() == [2, 0, 1] and r.tolist() == [[0, 1], [2, 3], [16, 17]]
        assert sh.all_gather_frames(torch.tensor([5, 6, 7]), 3).tolist() == [5, 6, 7]
    finally:
        sh.close()


def test_cascade_roundtrips_through_bytes():
    from grayskull_amd.cascade import Cascade
    path = os.path.join(ROOT, "tests", "golden", "frontalface_cascade.bin")
    raw = open(path, "rb").read()
    a, b = Cascade.from_blob(path), Cascade.from_bytes(raw)
    assert (a.nfeatures, a.nweaks, a.nstages) == (b.nfeatures, b.nweaks, b.nstages) == (136, 139, 20)
    assert bytes(a.subsets) == bytes(b.subsets) and bytes(a.stage_threshold) == bytes(b.stage_threshold)
    with pytest.raises(ValueError):
        Cascade.from_bytes(raw[:1000])
    with pytest.raises(ValueError):
        Cascade.from_bytes(b"nope")


def test_gsbatch_shards_files_like_the_python_driver():
    """the C99 driver's `--gpus N` split (grayskull_amd/host/gsbatch.c frame_range) is the rule bench.py uses
    (grayskull_amd/shard.py): same ranges for every (workers, files), each file owned exactly once"""
    import subprocess
    from grayskull_amd.shard import frame_range
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "tool"])
    exe = os.path.join(ROOT, "grayskull_amd", "gsbatch")
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 8, 9, 4096, 4099):
            out = subprocess.run([exe, "--shard-table", str(world), str(total)], capture_output=True, text=True, check=True).stdout
            got = [tuple(int(v) for v in line.split()) for line in out.splitlines()]
            assert got == [(r,) + frame_range(r, world, total) for r in range(world)], (world, total)
            assert [f for _, lo, hi in got for f in range(lo, hi)] == list(range(total))
