"""The reference's OWN unit tests (reference test.c, unmodified, read where it lies) compiled
against this repo's drop-in symbols: every hot-path function test.c calls (gs_blur, gs_histogram,
gs_threshold, gs_adaptive_threshold, gs_otsu_threshold, gs_erode/gs_dilate, gs_sobel, gs_integral,
gs_crop, gs_resize, gs_match_template, gs_find_best_match)
resolves to OUR implementation; everything out of scope (gs_crop, gs_resize, gs_blobs, ...) stays
the reference's static inline code, as SURVEY.md 8(b) prescribes.

On this CPU-only box "our implementation" is the kernel-logic emulator build of the same sources
(tests/emu/libgs_kernel_emu.so).  The same programs linked against libgrayskull_hip.so are prebuilt
by `make -C oracle ref` (oracle/_ref/ref_test_hip, nano_hip) and run on the GPU box by
tests/test_gpu_vs_reference.py, which has no /root/reference.
Nothing from the reference is copied: the wrapper #includes the files in place."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _ensure_emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "emu"])


def _wrapper(tmp_path, unit):
    """oracle/gen_ref_wrapper.py: the reference program, unmodified, with the hot-path symbols extern"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from gen_ref_wrapper import wrapper_text
    text, names = wrapper_text(REF, unit)
    src = tmp_path / "wrapper.c"
    src.write_text(text)
    return str(src), names


def _build_and_run(tmp_path, libdir, libname, unit, args=()):
    src, names = _wrapper(tmp_path, unit)
    exe = str(tmp_path / "ref_suite")
    cmd = ["gcc", "-std=c99", "-O1", "-w", "-I" + REF, src, "-o", exe, "-L" + libdir, "-l" + libname,
           "-Wl,-rpath," + libdir, "-lm"]
    subprocess.check_call(cmd)
    # the hot-path calls really bind to the shared library
    undefined = subprocess.check_output(["nm", "-u", exe]).decode()
    for n in ("gs_blur", "gs_sobel", "gs_integral", "gs_otsu_threshold"):
        assert re.search(r"\bU %s\b" % n, undefined), n + " did not bind to the drop-in library"
    return subprocess.run([exe, *args], capture_output=True, timeout=900)


@pytest.mark.skipif(not os.path.exists(REF + "/test.c"), reason="reference checkout not present")
def test_reference_unit_tests_pass_against_our_kernels_emulated(tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    _ensure_emu()
    r = _build_and_run(tmp_path, emu, "gs_kernel_emu", REF + "/test.c")
    assert r.returncode == 0, (r.stdout + r.stderr).decode()[-2000:]


NANO = REF + "/examples/nanomagick/nanomagick.c"
# verb, arguments -- a subset of the reference Makefile's `testdata` recipe (Makefile:10-33) plus the
# feature verbs it never exercises; all on the reference's own 128x128 fixture
NANO_RUNS = [("blur", ["1"]), ("blur", ["9"]), ("threshold", ["otsu"]), ("threshold", ["128"]),
             ("adaptive", ["15", "5"]), ("sobel", []), ("morph", ["erode", "2"]), ("morph", ["dilate", "2"]),
             ("keypoints", ["100", "20"]), ("faces", ["1"]), ("scan", []),
             ("resize", ["128", "64"]), ("resize", ["300", "200"]), ("crop", ["32", "32", "64", "64"])]


@pytest.mark.skipif(not os.path.exists(NANO), reason="reference checkout not present")
def test_nanomagick_cli_same_outputs_with_our_kernels_emulated(tmp_path):
    """the reference's CLI (unmodified) linked against our symbols writes the same PGM bytes and
    prints the same text as the reference's own build, verb by verb"""
    emu = os.path.join(ROOT, "tests", "emu")
    _ensure_emu()
    ref_exe = str(tmp_path / "nano_ref")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-w", "-I" + REF, "-I" + os.path.dirname(NANO), NANO,
                           "-o", ref_exe, "-lm"])
    src, _ = _wrapper(tmp_path, NANO)
    our_exe = str(tmp_path / "nano_ours")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-w", "-I" + REF, "-I" + os.path.dirname(NANO), src, "-o", our_exe,
                           "-L" + emu, "-lgs_kernel_emu", "-Wl,-rpath," + emu, "-lm"])
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    for i, (verb, args) in enumerate(NANO_RUNS):
        outs = []
        for tag, exe in (("ref", ref_exe), ("ours", our_exe)):
            out = str(tmp_path / ("%s_%d.pgm" % (tag, i)))
            r = subprocess.run([exe, verb, *args, lena, out], capture_output=True, timeout=900)
            assert r.returncode == 0, (verb, tag, r.stderr.decode()[-500:])
            outs.append((open(out, "rb").read() if os.path.exists(out) else b"", r.stdout))
        assert outs[0][0] == outs[1][0], "nanomagick %s %s: output image differs" % (verb, args)
        assert outs[0][1] == outs[1][1], "nanomagick %s %s: printed text differs" % (verb, args)
        assert outs[0][0] or outs[0][1], "nanomagick %s produced nothing to compare" % verb


@pytest.mark.skipif(not os.path.exists(REF + "/testdata/aruco.pgm"), reason="reference checkout not present")
def test_nanomagick_makefile_chain_on_aruco(tmp_path):
    """the multi-stage pipe of the reference Makefile's `testdata` recipe (Makefile:25-30) on its
    640x480 fixture: blur 3 | sobel | threshold otsu | morph dilate 9 | morph erode 10 | blobs 150"""
    emu = os.path.join(ROOT, "tests", "emu")
    _ensure_emu()
    inc = ["-I" + REF, "-I" + os.path.dirname(NANO)]
    ref_exe, our_exe = str(tmp_path / "nano_ref"), str(tmp_path / "nano_ours")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-w", *inc, NANO, "-o", ref_exe, "-lm"])
    src, _ = _wrapper(tmp_path, NANO)
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-w", *inc, src, "-o", our_exe, "-L" + emu, "-lgs_kernel_emu",
                           "-Wl,-rpath," + emu, "-lm"])
    chain = [("blur", ["3"]), ("sobel", []), ("threshold", ["otsu"]), ("morph", ["dilate", "9"]),
             ("morph", ["erode", "10"]), ("blobs", ["150"])]
    results = []
    for tag, exe in (("ref", ref_exe), ("ours", our_exe)):
        cur = REF + "/testdata/aruco.pgm"
        for i, (verb, args) in enumerate(chain):
            out = str(tmp_path / ("%s_chain_%d.pgm" % (tag, i)))
            r = subprocess.run([exe, verb, *args, cur, out], capture_output=True, timeout=900)
            assert r.returncode == 0, (verb, tag, r.stderr.decode()[-500:])
            cur = out
        results.append([open(str(tmp_path / ("%s_chain_%d.pgm" % (tag, i))), "rb").read() for i in range(len(chain))])
    for i, (verb, args) in enumerate(chain):
        assert results[0][i] == results[1][i], "stage %d (%s %s) differs" % (i, verb, args)
