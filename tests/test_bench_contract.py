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


def _walk(d, path=""):
    if isinstance(d, dict):
        for k, v in d.items():
            yield from _walk(v, path + "/" + str(k))
    elif isinstance(d, list):
        for i, v in enumerate(d):
            yield from _walk(v, path + "/%d" % i)
    else:
        yield path, d


def test_latest_committed_bench_line_honours_the_contract():
    """the newest profiles/r*_bench.json (what `python bench.py` printed on the GPU box): the driver's keys are there,
    the metric is BASELINE.json's, every fraction in the line is physical (<= 1), the dominant-kernel roofline carries
    achieved / peak / frac / traffic, the CPU baseline says how it was taken"""
    import glob
    import json
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json"))
                   if re.fullmatch(r"r\d\d[a-z]_bench\.json", os.path.basename(f)))  # the default workload's line only
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == base["metric"] and d["unit"] == "Mpix/s" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "u8" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_frames"] * 3840 * 2160 / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or 0.9 < r["traffic"] / (2 * 32 * 3840 * 2160) < 1.2  # PMC bytes per 32-frame launch vs 2 B/px
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    for path, v in _walk(d):
        leaf = path.rsplit("/", 1)[-1]
        # frac_of_copy_ceiling relates to the guide's measured 6.29 TB/s two-buffer copy (the in-place threshold stream
        # passes it); every other fraction is against a physical peak and cannot exceed 1
        if isinstance(v, (int, float)) and "frac" in leaf and "copy_ceiling" not in leaf and "percall" not in path:
            assert 0 <= v <= 1.0, (path, v)


def test_stamped_measurement_files_describe_the_kernels_in_the_tree():
    """profiles/*.json that bench.py reads carry the hashes of the kernel sources they were taken from
    (scripts/stamp.py).  fused_isa_mix.json can be regenerated without a GPU (make -C grayskull_amd/csrc asm +
    scripts/isa_count.py), so a stale one is an error; the PMC files need the GPU box -- bench.py reports their
    numbers as null while they are stale, and this test only checks that the mechanism sees the change."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from stamp import fresh
    mix = json.load(open(os.path.join(ROOT, "profiles", "fused_isa_mix.json")))
    assert fresh(mix) is True, "k_fused.h / gs_fused.cpp / k_strip.h changed after profiles/fused_isa_mix.json: regenerate it " \
                               "(make -C grayskull_amd/csrc asm; scripts/isa_count.py ...; scripts/stamp.py ...)"
    stale = dict(mix, kernel_sources={k: "0" * 40 for k in mix["kernel_sources"]})
    assert fresh(stale) is False and fresh({}) is None
    b = _bench()
    d, stamp = b.measurement(os.path.join(ROOT, "profiles", "fused_isa_mix.json"))
    assert d is not None and stamp["kernel_sources_unchanged"] is True
