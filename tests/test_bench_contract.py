"""bench.py's launch contract, checked without a GPU: `python bench.py --gpus N` on its own must become a
torchrun launch of N ranks on 127.0.0.1 (or refuse when fewer GPUs are visible); under the driver's own
torchrun launch (WORLD_SIZE set) it must not re-spawn."""
import argparse
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_n_spawns_n_ranks(monkeypatch):
    import torch
    b = _bench()
    calls = []
    monkeypatch.setattr(os, "execv", lambda exe, argv: calls.append((exe, argv)))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("GS_BENCH_BACKEND", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    b.spawn_ranks_if_needed(argparse.Namespace(gpus=8))
    assert len(calls) == 1
    exe, argv = calls[0]
    assert exe == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and argv[argv.index("--nproc-per-node") + 1] == "8"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    assert argv[-6:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"] and argv[-7].endswith("bench.py")


def test_gpus_n_refuses_a_smaller_machine(monkeypatch):
    import torch
    b = _bench()
    monkeypatch.setattr(os, "execv", lambda *a: pytest.fail("must not launch"))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("GS_BENCH_BACKEND", raising=False)
    with pytest.raises(SystemExit) as e:
        b.spawn_ranks_if_needed(argparse.Namespace(gpus=2))
    assert "only 1 GPU(s) visible" in str(e.value)


def test_no_respawn_under_torchrun_or_for_one_gpu(monkeypatch):
    b = _bench()
    monkeypatch.setattr(os, "execv", lambda *a: pytest.fail("must not launch"))
    monkeypatch.setenv("WORLD_SIZE", "8")
    b.spawn_ranks_if_needed(argparse.Namespace(gpus=8))  # one of the driver's ranks
    monkeypatch.delenv("WORLD_SIZE")
    b.spawn_ranks_if_needed(argparse.Namespace(gpus=1))
