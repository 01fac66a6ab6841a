"""The C boundary: headers compile as strict C99, struct layouts match the reference ABI,
the product library loads and exports every symbol the headers declare (no compute here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared_functions():
    names = set()
    for hdr in ("grayskull.h", "grayskull_hip.h"):
        txt = open(os.path.join(INC, hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"#ifdef GS_EXPERIMENT.*?#endif", "", txt, flags=re.S)  # not in the release library
        txt = "\n".join(ln for ln in txt.splitlines() if not ln.lstrip().startswith("#define"))
        for m in re.finditer(r"\b(gsh?_[a-z0-9_]+)\s*\(", txt):
            names.add(m.group(1))
    # header inlines / macros are not library symbols
    return names - {"gs_valid", "gs_get", "gs_set", "gs_integral_sum", "gs_for"}


def test_headers_are_strict_c99_and_layouts_match(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "grayskull_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu ", sizeof(struct gs_image), sizeof(struct gs_rect),
         sizeof(struct gs_point), sizeof(struct gs_keypoint), sizeof(struct gs_match),
         sizeof(struct gs_lbp_cascade));
  printf("%zu %zu %zu %zu %zu\n", offsetof(struct gs_image, data), offsetof(struct gs_keypoint, response),
         offsetof(struct gs_keypoint, angle), offsetof(struct gs_keypoint, descriptor),
         offsetof(struct gs_lbp_cascade, features));
  return 0;
}''')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", INC,
                           str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    # 16/16/8/48/12/96; data@8, response@8, angle@12, descriptor@16, features@16
    assert out == ["16", "16", "8", "48", "12", "96", "8", "8", "12", "16", "16"]


def test_library_loads_and_exports_every_declared_symbol():
    import grayskull_amd as G
    lib = G.lib()  # raises ImportError if libgrayskull_hip.so was not built
    assert "gfx950" in lib.version()
    declared = declared_functions()
    assert declared == set(G.EXPORTED_SYMBOLS), declared ^ set(G.EXPORTED_SYMBOLS)
    nm = subprocess.check_output(["nm", "-D", "--defined-only", G.HIP_LIBRARY]).decode()
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    missing = declared - exported
    assert not missing, "declared in include/*.h but not exported: %s" % sorted(missing)


def test_product_library_has_gfx950_code_and_no_oracle():
    import grayskull_amd as G
    blob = open(G.HIP_LIBRARY, "rb").read()
    assert b"gfx950" in blob
    nm = subprocess.check_output(["nm", "-D", G.HIP_LIBRARY]).decode()
    assert "orc_" not in nm and "emu" not in nm.lower().replace("hipmemu", "")


def test_c99_dropin_program_against_emulated_kernels(tmp_path, emu):
    """tests/c/test_dropin.c (C99, -pedantic) linked with the kernel-logic emulator build"""
    exe = tmp_path / "dropin"
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", INC,
                           os.path.join(ROOT, "tests", "c", "test_dropin.c"), "-o", str(exe),
                           "-L", emu_dir, "-l:libgs_kernel_emu.so", "-Wl,-rpath," + emu_dir])
    out = subprocess.check_output([str(exe)]).decode()
    assert "all passed" in out


def run_nostdlib_program(tmp_path, libdir, libname, frame, nkps, threshold):
    """tests/c/test_nostdlib.c built -std=c99 -pedantic -DGS_NO_STDLIB against the drop-in header, linked with `libname`;
    returns (keypoints, [(angle bits, descriptor)] of the separate single-keypoint calls)"""
    import numpy as np
    from grayskull_amd import KEYPOINT_DTYPE
    exe = tmp_path / "nostdlib"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-DGS_NO_STDLIB", "-I", INC,
                           os.path.join(ROOT, "tests", "c", "test_nostdlib.c"), "-o", str(exe),
                           "-L", libdir, "-l:" + libname, "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    # the object must reference the *_nostdlib symbols, and none of the libm-flavour ones
    nm = subprocess.check_output(["nm", "-u", str(exe)]).decode()
    undef = {ln.split()[-1].split("@")[0] for ln in nm.splitlines() if ln.strip()}
    assert {"gs_orb_extract_nostdlib", "gs_compute_orientation_nostdlib", "gs_brief_descriptor_nostdlib"} <= undef
    assert not ({"gs_orb_extract", "gs_compute_orientation", "gs_brief_descriptor"} & undef)
    h, w = frame.shape
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(np.array([w, h, nkps, threshold], np.uint32).tobytes())
        f.write(np.ascontiguousarray(frame, np.uint8).tobytes())
    subprocess.check_call([str(exe), str(fin), str(fout)])
    raw = open(fout, "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    kps = np.frombuffer(raw[4:4 + 48 * n], KEYPOINT_DTYPE).copy()
    tail = np.frombuffer(raw[4 + 48 * n:], np.uint32)
    rest, border = tail[:-4].reshape(-1, 9), [int(b) for b in tail[-4:]]
    return kps, [(int(r[0]), r[1:].copy()) for r in rest], border


def check_nostdlib_program(tmp_path, libdir, libname, ref_ns, frame, nkps, threshold=20):
    import numpy as np
    kps, singles, border = run_nostdlib_program(tmp_path, libdir, libname, frame, nkps, threshold)
    ko = ref_ns.orb_extract(frame, nkps, threshold)
    assert len(kps) == len(ko), "%d vs %d keypoints" % (len(kps), len(ko))
    assert kps.tobytes() == ko.tobytes(), "gs_orb_extract under -DGS_NO_STDLIB differs from the reference built the same way"
    assert len(singles) == min(len(ko), 8)
    for (abits, desc), k in zip(singles, ko):
        assert abits == int(np.float32(ref_ns.orientation(frame, int(k["x"]), int(k["y"]), 15)).view(np.uint32))
        assert abits == int(np.float32(k["angle"]).view(np.uint32))
        assert np.array_equal(desc, ref_ns.brief(frame, int(k["x"]), int(k["y"]), float(k["angle"])))
        assert np.array_equal(desc, k["desc"])
    # keypoints closer than r to the border: no assert under GS_NO_STDLIB (ref :69), out-of-image pixels read 0 (ref :41-43)
    h, w = frame.shape
    for abits, (x, y) in zip(border, [(2, 3), (w - 1, h - 1), (0, 0), (w - 5, 7)]):
        assert abits == int(np.float32(ref_ns.orientation(frame, x, y, 15)).view(np.uint32)), (x, y)


def test_c99_nostdlib_caller_gets_the_polynomial_flavour(tmp_path, emu):
    """the GS_NO_STDLIB seam of include/grayskull.h (ref :68-101; examples/wasm/grayskull.c:31-35) against the emulator
    build: a C99 caller compiled -DGS_NO_STDLIB equals the reference header compiled -DGS_NO_STDLIB, bit for bit --
    and differs from the libm flavour, or the seam would prove nothing"""
    import numpy as np
    from oracle import pyoracle
    from oracle.pyoracle import Oracle
    if not pyoracle.have_reference_nostdlib():
        pytest.skip("oracle/_ref/libgs_ref_nostdlib.so not present")
    ref_ns = Oracle("reference_nostdlib")
    emu_dir = os.path.join(ROOT, "tests", "emu")
    frame = Oracle.synth(320, 240, 5)
    check_nostdlib_program(tmp_path, emu_dir, "libgs_kernel_emu.so", ref_ns, frame, 120)
    libm = Oracle("port").orb_extract(frame, 120, 20)
    ns = ref_ns.orb_extract(frame, 120, 20)
    assert len(libm) == len(ns) > 8 and libm.tobytes() != ns.tobytes()


def test_blur_magic_divisions_are_exact():
    """(s*MUL) >> 24 == s // d over the whole reachable range, product < 2^32, MUL < 2^24
    (k_stencil.h BlurMagic: the quotient is the top byte of a v_mul_u32_u24 product)"""
    for r, mul in ((1, 1864136), (2, 671089), (3, 342393)):
        d = (2 * r + 1) ** 2
        assert mul < 2 ** 24
        for s in range(0, 255 * d + 1):
            assert (s * mul) >> 24 == s // d
            assert s * mul < 2 ** 32


def test_precondition_failures_abort_like_gs_assert(emu, tmp_path):
    """reference behaviour (grayskull.h:94-98): message 'Assertion failed: <cond>' on stderr + abort()"""
    import sys
    prog = tmp_path / "bad.py"
    prog.write_text('''
import sys, numpy as np
sys.path.insert(0, %r)
import grayskull_amd as G
g = G.Grayskull(%r)
g.blur(np.zeros((4, 4), np.uint8), np.zeros((5, 4), np.uint8), 1)   # dst/src size mismatch
''' % (ROOT, os.path.join(ROOT, "tests", "emu", "libgs_kernel_emu.so")))
    r = subprocess.run([sys.executable, str(prog)], capture_output=True)
    assert r.returncode == -6, r  # SIGABRT
    assert b"Assertion failed:" in r.stderr and b"dst.w == src.w" in r.stderr


def test_filter_strip_kernel_quotient_is_exact():
    """k_filter16 divides the clamped non-negative sum c <= min(255*norm, 32767) by norm as the top
    byte of c * ceil(2^24 / norm); check every (norm <= 256, c) pair against integer division"""
    import numpy as np
    for norm in range(1, 257):
        mul = ((1 << 24) + norm - 1) // norm
        cap = min(255 * norm, 32767)
        c = np.arange(cap + 1, dtype=np.uint64)
        prod = c * np.uint64(mul)
        assert int(prod.max()) < (1 << 32), norm
        assert np.array_equal(prod >> np.uint64(24), np.minimum(c // np.uint64(norm), 255)), norm
