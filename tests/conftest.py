"""pytest configuration.

Markers:
  gpu  -- needs a real MI355X (run on the GPU box: `pytest -m gpu`).  Everything else runs on
          CPU: oracle vs golden vectors, ABI/symbol checks, gloo sharding, and the kernel
          sources exercised through the host-fiber emulator (tests/emu, a test tool).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_SO = os.path.join(ROOT, "tests", "emu", "libgs_kernel_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD MI355X GPU (run with -m gpu on the GPU box)")


def _make(target):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), target])


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle("port")


@pytest.fixture(scope="session")
def reference():
    """the compiled, unmodified reference (oracle/_ref); skipped where it was never built"""
    from oracle import pyoracle
    if not pyoracle.have_reference():
        if os.path.exists("/root/reference/grayskull.h"):
            pyoracle.build()
        else:
            pytest.skip("oracle/_ref/libgs_ref.so not present")
    return pyoracle.Oracle("reference")


@pytest.fixture(scope="session")
def cascade():
    from grayskull_amd.cascade import Cascade
    return Cascade.from_blob(os.path.join(GOLDEN, "frontalface_cascade.bin"))


@pytest.fixture(scope="session")
def emu():
    """kernel sources compiled for the host-fiber emulator (logic check without a GPU)"""
    import grayskull_amd as G
    _make("emu")
    return G.Grayskull(EMU_SO)


@pytest.fixture(scope="session")
def hip():
    """the product library on a real GPU"""
    import grayskull_amd as G
    g = G.lib()
    assert g.device_count() > 0, "no HIP device visible"
    # GS_TUNE_KEY / GS_TUNE_VAL: run the GPU suite under a non-default launch-tuning key (results never change)
    if os.environ.get("GS_TUNE_KEY"):
        g.tune(int(os.environ["GS_TUNE_KEY"]), int(os.environ.get("GS_TUNE_VAL", "0")))
    return g


@pytest.fixture(scope="session")
def kat():
    import json
    return json.load(open(os.path.join(GOLDEN, "kat.json")))
