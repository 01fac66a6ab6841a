"""Differential run on the GPU box: the HIP library against the COMPILED, UNMODIFIED reference
(oracle/_ref/libgs_ref.so -- built in the build container by oracle/Makefile from /root/reference where
it lies, shipped as a git-ignored binary) instead of against the restatement, on adversarial
shapes: strip-kernel boundaries (w multiple of 16 / 1024 +- 1), tiny heights, radii >= height,
extreme values, cap edges.  Plus the reference's own programs -- test.c and the nanomagick CLI,
unmodified -- linked against libgrayskull_hip.so (prebuilt by `make -C oracle ref`, since
/root/reference does not exist on the GPU box)."""
import os
import subprocess

import numpy as np
import pytest

import parity_cases as pc
from oracle.pyoracle import Oracle
from util import assert_same, random_cascade

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
HOST, DEV = pc.Mem("host"), pc.Mem("device")

# the reference's blur / adaptive threshold cost (2r+1)^2 bounds-checked taps per pixel on one host core: keep
# radius x pixels small enough that the whole module takes about a minute of CPU time
SHAPES = [(1024, 9), (1023, 9), (1025, 9), (1040, 33), (4096, 5), (3840, 24), (48, 48), (17, 3), (3, 17),
          (7, 7), (16, 2), (2048, 3), (131, 77), (800, 600)]


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", SHAPES)
def test_stencils_pointwise_integral_vs_reference(hip, reference, shape, mem):
    w, h = shape
    for img in (Oracle.synth(w, h, 31 * w + h), np.random.RandomState(w * h).randint(0, 256, (h, w)).astype(np.uint8)):
        pc.stencils(hip, reference, img, mem, radii=(1, 2, 3, 7, max(h, 4)) if w * h < 50000 else (1, 2, 3))
        pc.pointwise(hip, reference, img, mem)
        pc.integral(hip, reference, img, mem)
        pc.next_rows(hip, reference, img, mem)


def test_extreme_values_vs_reference(hip, reference):
    for v in (0, 1, 254, 255):
        img = np.full((40, 1056), v, np.uint8)
        pc.stencils(hip, reference, img, DEV)
        pc.pointwise(hip, reference, img, DEV)
    chk = ((np.indices((64, 2048)).sum(0) % 2) * 255).astype(np.uint8)
    pc.stencils(hip, reference, chk, DEV, radii=(1, 2, 3))
    stripes = np.zeros((48, 1040), np.uint8)
    stripes[:, ::3] = 255
    pc.stencils(hip, reference, stripes, DEV, radii=(1, 2, 3))
    pc.pointwise(hip, reference, stripes, HOST)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", [(67, 45), (131, 77), (640, 480), (1280, 720), (1031, 64)])
def test_fast_orb_vs_reference(hip, reference, shape, mem):
    w, h = shape
    img = Oracle.synth(w, h, 5 * w + h)
    pc.fast(hip, reference, img, mem, threshold=20, caps=(5000, 7, 1))
    pc.fast(hip, reference, np.random.RandomState(w).randint(0, 256, (h, w)).astype(np.uint8), mem, threshold=40, caps=(300,))
    pc.orb(hip, reference, img, mem, nkps=50 if w < 600 else 500)
    pc.fast_unsigned_wrap_quirk(hip, reference, mem)


def test_geometry_and_pyramid_vs_reference(hip, reference):
    img = Oracle.synth(320, 240, 12)
    pc.geometry(hip, reference, img, HOST)
    pc.geometry(hip, reference, img, DEV)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
def test_lbp_vs_reference(hip, reference, cascade, mem):
    pc.lbp(hip, reference, Oracle.synth(200, 160, 7), mem, cascade,
           params=((4096, 1.1, 1.0, 4.0, 1), (10, 1.3, 1.0, 2.0, 3), (1, 1.1, 1.0, 4.0, 1)),
           windows=((0, 0, 1.0), (5, 3, 1.2), (176, 136, 1.0), (177, 136, 1.0), (0, 0, 3.4)))
    edges = reference.sobel(reference.blur(Oracle.synth(640, 360, 1000), 2))
    pc.lbp(hip, reference, edges, mem, cascade, params=((4096, 1.1, 1.0, 4.0, 1), (50, 1.1, 1.0, 4.0, 1), (3, 1.2, 1.0, 3.0, 2)))
    pc.lbp(hip, reference, Oracle.synth(320, 200, 9), mem, random_cascade(1),
           params=((4096, 1.25, 1.0, 2.0, 2), (37, 1.25, 1.0, 2.0, 1), (100000, 1.2, 1.0, 3.0, 1)))


# ---- the reference's own programs against the HIP library (prebuilt where /root/reference exists) ----
def _prebuilt(name):
    p = os.path.join(REFDIR, name)
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/%s was not prebuilt (make -C oracle ref in the build container)" % name)
    return p


def test_reference_test_c_passes_against_hip_library():
    """reference test.c (unmodified; ref test.c:351-366 runs its 13 tests) with every hot-path symbol
    bound to libgrayskull_hip.so"""
    exe = _prebuilt("ref_test_hip")
    und = subprocess.check_output(["nm", "-u", exe]).decode()
    for n in ("gs_blur", "gs_sobel", "gs_integral", "gs_otsu_threshold"):
        assert (" U " + n) in und, n + " is not bound to the drop-in library"
    r = subprocess.run([exe], capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr).decode()[-2000:]


NANO_RUNS = [("blur", ["1"]), ("blur", ["9"]), ("threshold", ["otsu"]), ("threshold", ["128"]),
             ("adaptive", ["15", "5"]), ("sobel", []), ("morph", ["erode", "2"]), ("morph", ["dilate", "2"]),
             ("keypoints", ["100", "20"]), ("orb", ["<lena>"]), ("faces", ["1"]), ("scan", []),
             ("resize", ["128", "64"]), ("resize", ["300", "200"]), ("crop", ["32", "32", "64", "64"])]


def test_nanomagick_cli_same_outputs_on_hip_library(tmp_path):
    """the reference CLI (unmodified) linked against the HIP library writes the same PGM bytes and
    prints the same text as the reference's own build, verb by verb"""
    ref_exe, our_exe = _prebuilt("nano_ref"), _prebuilt("nano_hip")
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    for i, (verb, args) in enumerate(NANO_RUNS):
        outs = []
        for tag, exe in (("ref", ref_exe), ("ours", our_exe)):
            out = str(tmp_path / ("%s_%d.pgm" % (tag, i)))
            a = [lena if x == "<lena>" else x for x in args]  # `orb <template.pgm>` (nanomagick.c:292): lena against itself
            r = subprocess.run([exe, verb, *a, lena, out], capture_output=True, timeout=600)
            assert r.returncode == 0, (verb, tag, r.stderr.decode()[-500:])
            outs.append((open(out, "rb").read() if os.path.exists(out) else b"", r.stdout))
        assert outs[0][0] == outs[1][0], "nanomagick %s %s: output image differs" % (verb, args)
        assert outs[0][1] == outs[1][1], "nanomagick %s %s: printed text differs" % (verb, args)


def test_device_resident_orb_nostdlib_at_the_kat_sizes(hip):
    """gsh_orb_extract_batch_nostdlib (no host round trip) == the reference header compiled with -DGS_NO_STDLIB
    (oracle/_ref/libgs_ref_nostdlib.so) at the three KAT sizes of SURVEY 8(c), nkps = 500, threshold 20"""
    from oracle import pyoracle
    if not pyoracle.have_reference_nostdlib():
        pytest.skip("oracle/_ref/libgs_ref_nostdlib.so was not prebuilt")
    ref_ns = Oracle("reference_nostdlib")
    for (w, h, seed, nf) in ((1920, 1080, 3, 2), (1280, 720, 4, 3), (67, 45, 5, 2), (640, 480, 11, 4)):
        frames = np.stack([Oracle.synth(w, h, seed + 100 * i) for i in range(nf)])
        pc.orb_nostdlib(hip, ref_ns, frames, nkps=500)
    pc.orb_nostdlib(hip, ref_ns, np.stack([Oracle.synth(320, 240, 5)]), nkps=2000)  # cap 5000
    pc.orb_nostdlib(hip, ref_ns, np.full((2, 64, 64), 77, np.uint8), nkps=50)       # nothing to find


def test_fast_score_map_of_another_size_vs_reference(hip, reference):
    from test_emu_logic import test_fast_with_a_score_map_of_another_size as body
    body(hip, reference)


def test_c99_caller_built_with_gs_no_stdlib_gets_the_polynomial_flavour(hip, tmp_path):
    """the GS_NO_STDLIB seam of include/grayskull.h (ref :68-101; examples/wasm/grayskull.c:31-35) through the C ABI on the
    MI355X: tests/c/test_nostdlib.c compiled -std=c99 -pedantic -DGS_NO_STDLIB and linked against libgrayskull_hip.so
    equals oracle/_ref/libgs_ref_nostdlib.so (the reference header compiled with the same macro) bit for bit -- keypoints,
    angles, descriptors of gs_orb_extract and of separate gs_compute_orientation / gs_brief_descriptor calls"""
    from oracle import pyoracle
    import grayskull_amd as G
    from test_abi import check_nostdlib_program
    if not pyoracle.have_reference_nostdlib():
        pytest.skip("oracle/_ref/libgs_ref_nostdlib.so was not prebuilt")
    ref_ns = Oracle("reference_nostdlib")
    libdir, libname = os.path.split(G.HIP_LIBRARY)
    for i, (w, h, seed, nkps) in enumerate(((1280, 720, 4, 500), (320, 240, 5, 2000), (67, 45, 5, 500))):
        d = tmp_path / ("case%d" % i)
        d.mkdir()
        check_nostdlib_program(d, libdir, libname, ref_ns, Oracle.synth(w, h, seed), nkps)

