"""RCCL on the hardware that is there: bench.py under `torch.distributed.run` with ONE rank and
GS_BENCH_FORCE_DIST=1, so that the process group is nccl (== RCCL on ROCm) and every collective of the N > 1
path -- barrier, all_reduce, all_gather on device tensors, the cascade broadcast, the packed rect gather --
really executes on the GPU; and every frame of the batch is compared with the reference-generated golden
checksums (tests/golden/batch_checksums.json)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun_bench(*args):
    env = dict(os.environ, GS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GS_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + list(args)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_cfg1_under_rccl_world1_checks_every_frame():
    d = _torchrun_bench("--frames", "64", "--steps", "2", "--warmup", "1", "--no-other", "--no-cpu")
    assert d["backend"] == "nccl" and d["rccl_ranks_seen"] == 1 and d["n_gpus"] == 1
    assert "MISMATCH" not in d["parity"], d["parity"]
    assert d["parity_golden"].startswith("64/64 checksums"), d["parity_golden"]
    assert d["otsu_thresholds_gathered"] == 64 and d["output_checksums_gathered"] == 64


def test_bench_cfg4_under_rccl_world1_broadcast_and_rect_gather():
    d = _torchrun_bench("--workload", "cfg4", "--frames", "3", "--steps", "1", "--warmup", "0")
    assert d["backend"] == "nccl" and d["rccl_ranks_seen"] == 1
    assert "MISMATCH" not in d["parity"] and "3 of the 3 frames" in d["parity"] and "frames 0-2" in d["parity"], d["parity"]
    assert d["detections_total"] > 0 and len(d["detections_first_frames"]) == 3


def test_sharder_collectives_on_device_tensors_over_nccl():
    """the Sharder's own calls with cuda tensors over a 1-rank nccl group, in a fresh process"""
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ["GS_BENCH_FORCE_DIST"] = "1"
from grayskull_amd.shard import Sharder
torch.cuda.set_device(0)
sh = Sharder()
assert sh.backend == "nccl" and sh.dist.get_backend() == "nccl"
sh.barrier()
assert sh.ranks_seen() == 1 and sh.max_over_ranks(2.5) == 2.5 and sh.min_over_ranks(0.0) == 0.0
assert sh.broadcast_bytes(b"LBPC" + bytes(range(100))) == b"LBPC" + bytes(range(100))
v = torch.arange(7, dtype=torch.int64, device="cuda")
assert sh.all_gather_frames(v, 7).tolist() == list(range(7))
rec = torch.arange(3 * 4 * 4, dtype=torch.int32, device="cuda").reshape(3, 4, 4)
c, r = sh.gather_varlen(torch.tensor([1, 0, 4], dtype=torch.int32, device="cuda"), rec, 3)
assert c.tolist() == [1, 0, 4] and r.is_cuda and r.tolist() == rec[0, :1].tolist() + rec[2].tolist()
sh.close()
print("nccl-world1-ok")
''' % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl-world1-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
