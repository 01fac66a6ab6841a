"""gsbatch (grayskull_amd/host/gsbatch.c): the multi-stage, multi-file PGM driver of SURVEY.md 8(f)
rank 3.  Its outputs must be byte-identical to piping the reference's nanomagick CLI through the same
verbs, file by file (reference Makefile:10-33, nanomagick.c:380-446).

CPU suite: the C99 program is linked with the kernel-logic emulator and compared with the
reference's own nanomagick build (compiled where it lies).  GPU suite: linked with
libgrayskull_hip.so and compared with the oracle's restatement of each verb (no /root/reference on
the GPU box)."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import read_pgm, assert_same

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
NANO = REF + "/examples/nanomagick/nanomagick.c"
SRC = os.path.join(ROOT, "grayskull_amd", "host", "gsbatch.c")
CFLAGS = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include")]


def write_pgm(path, a):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a).tobytes())


def build_emu(tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "emu"])
    exe = str(tmp_path / "gsbatch_emu")
    subprocess.check_call(CFLAGS + [SRC, "-o", exe, "-L" + emu, "-lgs_kernel_emu", "-pthread", "-Wl,-rpath," + emu])
    return exe


def build_ref_nano(tmp_path):
    exe = str(tmp_path / "nano_ref")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-w", "-I" + REF, "-I" + os.path.dirname(NANO), NANO, "-o", exe, "-lm"])
    return exe


def chain_args(chain):
    out = []
    for i, (verb, args) in enumerate(chain):
        if i:
            out.append(":")
        out += [verb, *args]
    return out


def nano_chain(exe, chain, src, tmp_path, tag):
    """the reference way: one process per verb per file; returns (final path or None, stderr text)"""
    cur, err = src, b""
    for i, (verb, args) in enumerate(chain):
        out = str(tmp_path / ("%s_%d.pgm" % (tag, i)))
        r = subprocess.run([exe, verb, *args, cur, out], capture_output=True, timeout=900)
        err += r.stderr
        if r.returncode != 0:
            return None, err
        cur = out
    return cur, err


CHAINS = [
    [("blur", ["3"]), ("sobel", []), ("threshold", ["otsu"]), ("morph", ["dilate", "2"]), ("morph", ["erode", "3"])],  # fused 3
    [("blur", ["2"]), ("sobel", [])],                                          # fused 2
    [("blur", ["9"]), ("sobel", []), ("threshold", ["40"])],                   # unfused (radius > 3)
    [("resize", ["200", "120"]), ("adaptive", ["7", "3"]), ("crop", ["8", "4", "96", "64"])],
    [("crop", ["1", "2", "100", "90"]), ("blur", ["1"]), ("threshold", ["otsu"])],
    [("sobel", []), ("morph", ["dilate", "1"]), ("resize", ["333", "77"])],
    [("threshold", ["300"])],                                                  # (uint8_t)300 = 44, like nanomagick
]


@pytest.mark.skipif(not os.path.exists(NANO), reason="reference checkout not present")
def test_gsbatch_equals_piped_nanomagick_emulated(tmp_path):
    exe, nano = build_emu(tmp_path), build_ref_nano(tmp_path)
    files = [os.path.join(ROOT, "tests", "golden", "lena.pgm"), REF + "/testdata/aruco.pgm"]
    from oracle.pyoracle import Oracle
    for k, (w, h) in enumerate([(160, 96), (160, 96), (131, 101)]):     # two frames share a group
        p = str(tmp_path / ("synth%d.pgm" % k))
        write_pgm(p, Oracle.synth(w, h, 20 + k))
        files.append(p)
    for c, chain in enumerate(CHAINS):
        outdir = tmp_path / ("out%d" % c)
        outdir.mkdir()
        r = subprocess.run([exe, "-v", "-o", str(outdir), *chain_args(chain), "--", *files], capture_output=True, timeout=1800)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        for i, f in enumerate(files):
            exp, _ = nano_chain(nano, chain, f, tmp_path, "ref%d_%d" % (c, i))
            got = open(str(outdir / os.path.basename(f)), "rb").read()
            assert exp is not None and got == open(exp, "rb").read(), "chain %d file %s differs" % (c, f)


CASCADE = os.path.join(ROOT, "tests", "golden", "frontalface_cascade.bin")
FEATURE_CHAINS = [
    [("keypoints", ["100", "20"])],
    [("blur", ["1"]), ("keypoints", ["30", "10"])],
    [("faces", ["1"])],
    [("blur", ["2"]), ("sobel", []), ("faces", ["1"])],   # the chain of BASELINE configs[4], through the CLI
    [("resize", ["200", "150"]), ("faces", ["2"])],
]


def read_records(path):
    return [tuple(int(x) for x in line.split()) for line in open(path).read().splitlines()]


def check_feature_outputs(outdir, f, chain, o, casc):
    """the sidecar records = what the reference computes before drawing (nanomagick.c:229-233, :362-364)"""
    img = oracle_chain(o, read_pgm(f), chain[:-1])
    verb, args = chain[-1]
    if verb == "keypoints":
        n, t = int(args[0]), int(args[1])
        kps, _ = o.fast(img, 5000, t)
        rec = read_records(str(outdir / (os.path.basename(f) + ".keypoints.txt")))
        assert len(rec) == min(n, len(kps))
        # qsort by response: the multiset of the strongest responses is fixed, their coordinates are among the detections
        assert sorted(r[2] for r in rec) == sorted(sorted((int(k) for k in kps["response"]), reverse=True)[:len(rec)])
        have = {(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps}
        assert all(r in have for r in rec)
    else:
        ro = o.lbp_detect(casc, o.integral(img), 100, 1.2, 1.0, 4.0, int(args[0]))
        rec = read_records(str(outdir / (os.path.basename(f) + ".faces.txt")))
        assert rec == [(int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])) for r in ro]


@pytest.mark.skipif(not os.path.exists(NANO), reason="reference checkout not present")
@pytest.mark.parametrize("gpus", [1, 2])
def test_gsbatch_feature_verbs_equal_piped_nanomagick_emulated(tmp_path, gpus):
    """keypoints / faces as the last stage: the drawn PGM is byte-identical to the piped reference CLI, the
    sidecar lists the detections; with --gpus 2 (two emulated devices, one host thread each) the files of a
    group are split over the workers and every output is still the same"""
    from oracle.pyoracle import Oracle
    from grayskull_amd.cascade import Cascade
    exe, nano = build_emu(tmp_path), build_ref_nano(tmp_path)
    o, casc = Oracle("port"), Cascade.from_blob(CASCADE)
    files = [os.path.join(ROOT, "tests", "golden", "lena.pgm")]
    for k, (w, h) in enumerate([(160, 120), (160, 120), (160, 120), (131, 101)]):   # three frames share a group
        p = str(tmp_path / ("synth%d.pgm" % k))
        a = Oracle.synth(w, h, 40 + k)
        a[0, 0] = max(int(a[0, 0]), 33)  # a raster that starts with a whitespace byte is unreadable for the reference (grayskull.h:116)
        write_pgm(p, a)
        files.append(p)
    env = dict(os.environ, GS_EMU_DEVICES="2")
    for c, chain in enumerate(FEATURE_CHAINS):
        outdir = tmp_path / ("fout%d" % c)
        outdir.mkdir()
        r = subprocess.run([exe, "-v", "--gpus", str(gpus), "--cascade", CASCADE, "-o", str(outdir), *chain_args(chain), "--", *files],
                           capture_output=True, timeout=1800, env=env)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        if gpus == 2:
            assert b"gpu 1 group 1: " in r.stderr and b"gpu 0 group 1: " in r.stderr  # both workers took a share of the 3-frame group
        compared = 0
        for i, f in enumerate(files):
            exp, err = nano_chain(nano, chain, f, tmp_path, "fref%d_%d" % (c, i))
            got = open(str(outdir / os.path.basename(f)), "rb").read()
            if exp is None:  # an intermediate PGM of the pipe started with a whitespace-valued pixel: the
                assert b"Could not load" in err, err  # reference cannot read its own output back (grayskull.h:116)
            else:
                assert got == open(exp, "rb").read(), "chain %d file %s: drawn image differs" % (c, f)
                compared += 1
            check_feature_outputs(outdir, f, chain, o, casc)
        assert compared >= len(files) - 1


@pytest.mark.skipif(not os.path.exists(NANO), reason="reference checkout not present")
def test_gsbatch_orb_verb_equals_nanomagick_emulated(tmp_path):
    """`orb <template.pgm>` as the last stage: the stitched picture with the 15 best matches is byte-identical to the
    reference CLI's, the sidecar's first line is the line nanomagick prints (nanomagick.c:308)"""
    from oracle.pyoracle import Oracle
    exe, nano = build_emu(tmp_path), build_ref_nano(tmp_path)
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    shifted = str(tmp_path / "shifted.pgm")
    a = read_pgm(lena)
    b = np.zeros_like(a)
    b[:-3, :-5] = a[3:, 5:]
    b[0, 0] = 40
    write_pgm(shifted, b)
    other = str(tmp_path / "synth.pgm")
    c = Oracle.synth(160, 120, 77)
    c[0, 0] = max(int(c[0, 0]), 33)
    write_pgm(other, c)
    files = [lena, shifted, other]
    for ci, chain in enumerate([[("orb", [lena])], [("blur", ["1"]), ("orb", [lena])]]):
        outdir = tmp_path / ("orb%d" % ci)
        outdir.mkdir()
        r = subprocess.run([exe, "-o", str(outdir), *chain_args(chain), "--", *files], capture_output=True, timeout=1800)
        for i, f in enumerate(files):
            cur, line = f, b""
            for k, (verb, args) in enumerate(chain):  # the reference way, keeping the last stage's stdout
                out = str(tmp_path / ("oref%d_%d_%d.pgm" % (ci, i, k)))
                rr = subprocess.run([nano, verb, *args, cur, out], capture_output=True, timeout=900)
                cur, line = (out if rr.returncode == 0 else None), rr.stdout
                if cur is None:
                    break
            rec = open(str(outdir / (os.path.basename(f) + ".orb.txt"))).read().splitlines()
            assert rec[0] + "\n" == line.decode(), (f, rec[0], line)
            mine = str(outdir / os.path.basename(f))
            if cur is None:  # no matches: nanomagick writes nothing, neither do we
                assert not os.path.exists(mine) and r.returncode == 1
            else:
                assert open(mine, "rb").read() == open(cur, "rb").read(), "orb picture differs for %s" % f
                nmatch = int(rec[0].rsplit(" ", 1)[1])
                assert len(rec) == 1 + nmatch
                d = [int(x.split()[2]) for x in rec[1:]]
                assert d == sorted(d)
    # an unreadable template: nanomagick's message, no output
    outdir = tmp_path / "orbbad"
    outdir.mkdir()
    r = subprocess.run([exe, "-o", str(outdir), "orb", str(tmp_path / "nope.pgm"), "--", lena], capture_output=True, timeout=600)
    assert r.returncode == 1 and b"Cannot load template image" in r.stdout and os.listdir(str(outdir)) == []


def test_gsbatch_feature_verb_errors_emulated(tmp_path):
    exe = build_emu(tmp_path)
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    out = tmp_path / "o"
    out.mkdir()
    r = subprocess.run([exe, "-o", str(out), "keypoints", "10", "20", ":", "sobel", "--", lena], capture_output=True)
    assert r.returncode == 1 and b"must be the last stage" in r.stderr
    r = subprocess.run([exe, "-o", str(out), "faces", "1", "--", lena], capture_output=True,
                       env={k: v for k, v in os.environ.items() if k != "GSBATCH_CASCADE"})
    assert r.returncode == 1 and b"faces needs a cascade blob" in r.stderr
    r = subprocess.run([exe, "--cascade", CASCADE, "-o", str(out), "faces", "0", "--", lena], capture_output=True)
    assert r.returncode == 1 and b"minimum neighbors must be positive" in r.stderr
    r = subprocess.run([exe, "-o", str(out), "keypoints", "0", "20", "--", lena], capture_output=True)
    assert r.returncode == 1 and b"Invalid number of keypoints or threshold" in r.stderr
    r = subprocess.run([exe, "--gpus", "3", "-o", str(out), "sobel", "--", lena], capture_output=True)
    assert r.returncode == 1 and b"--gpus 3 but only 1 HIP device(s) visible" in r.stderr
    assert os.listdir(str(out)) == []


@pytest.mark.skipif(not os.path.exists(NANO), reason="reference checkout not present")
def test_gsbatch_errors_like_nanomagick_emulated(tmp_path):
    """argument validation and per-file failures: same wording, exit 1, nothing written for the file"""
    exe, nano = build_emu(tmp_path), build_ref_nano(tmp_path)
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    black = str(tmp_path / "black.pgm")
    write_pgm(black, np.zeros((40, 48), np.uint8))
    bad = [[("blur", ["0"])], [("threshold", ["0"])], [("adaptive", ["3", "-1"])], [("morph", ["open", "2"])],
           [("morph", ["erode", "0"])], [("resize", ["0", "10"])], [("crop", ["100", "100", "64", "64"])]]
    for c, chain in enumerate(bad):
        outdir = tmp_path / ("bad%d" % c)
        outdir.mkdir()
        r = subprocess.run([exe, "-o", str(outdir), *chain_args(chain), "--", lena], capture_output=True, timeout=600)
        exp, err = nano_chain(nano, chain, lena, tmp_path, "badref%d" % c)
        assert exp is None and r.returncode == 1
        first = err.decode().splitlines()[0]
        assert first in r.stderr.decode(), (first, r.stderr.decode())
        assert os.listdir(str(outdir)) == []
    # an all-black frame has Otsu threshold 0: nanomagick refuses it, the other file still goes through
    outdir = tmp_path / "otsu0"
    outdir.mkdir()
    r = subprocess.run([exe, "-o", str(outdir), "threshold", "otsu", "--", black, lena], capture_output=True, timeout=600)
    exp, err = nano_chain(nano, [("threshold", ["otsu"])], black, tmp_path, "blackref")
    assert exp is None and r.returncode == 1 and err.decode().splitlines()[0] in r.stderr.decode()
    assert os.listdir(str(outdir)) == ["lena.pgm"]
    exp, _ = nano_chain(nano, [("threshold", ["otsu"])], lena, tmp_path, "lenaref")
    assert open(str(outdir / "lena.pgm"), "rb").read() == open(exp, "rb").read()
    # unreadable input, unknown verb, missing arguments
    r = subprocess.run([exe, "-o", str(outdir), "sobel", "--", str(tmp_path / "nope.pgm")], capture_output=True)
    assert r.returncode == 1 and b"Could not load" in r.stderr
    r = subprocess.run([exe, "-o", str(outdir), "sharpen", "--", lena], capture_output=True)
    assert r.returncode == 1 and b"Unknown command 'sharpen'" in r.stderr
    r = subprocess.run([exe, "-o", str(outdir), "crop", "1", "2", ":", "sobel", "--", lena], capture_output=True)
    assert r.returncode == 1 and b"Wrong number of arguments for 'crop'" in r.stderr


def test_gsbatch_pgm_reader_and_slicing_emulated(tmp_path):
    """(a) a raster whose first bytes are whitespace values: the reference's fscanf swallows them and
    rejects the file (grayskull.h:116), gsbatch loads it; (b) extra whitespace after maxval: raster
    starts after it, like the reference; (c) groups larger than the staging slice are cut into slices"""
    from oracle.pyoracle import Oracle
    exe = build_emu(tmp_path)
    o = Oracle("port")
    rs = np.random.RandomState(3)
    imgs = [rs.randint(0, 256, (24, 40)).astype(np.uint8) for _ in range(7)]
    imgs[0][0, :3] = (10, 32, 9)                    # newline, space, tab as the first pixels
    files = []
    for k, a in enumerate(imgs):
        p = str(tmp_path / ("r%d.pgm" % k))
        with open(p, "wb") as f:
            f.write(b"P5\n40 24\n255\n" + (b" \n" if k == 1 else b"") + a.tobytes())
        files.append(p)
    imgs[1][0, 0] = max(int(imgs[1][0, 0]), 33)    # make sure the raster of file 1 does not start with whitespace
    with open(files[1], "wb") as f:
        f.write(b"P5\n40 24\n255\n \n" + imgs[1].tobytes())
    if os.path.exists(NANO):
        nano = build_ref_nano(tmp_path)
        ref0, _ = nano_chain(nano, [("blur", ["1"])], files[0], tmp_path, "ws0")
        assert ref0 is None                         # the reference cannot read file 0 at all
        ref1, _ = nano_chain(nano, [("blur", ["1"])], files[1], tmp_path, "ws1")
        assert_same(read_pgm(ref1), o.blur(imgs[1], 1), "reference reads past the extra whitespace")
    outdir = tmp_path / "out"
    outdir.mkdir()
    env = dict(os.environ, GSBATCH_SLICE_BYTES=str(40 * 24 * 3))  # 3 frames per slice -> 3 slices
    r = subprocess.run([exe, "-v", "-o", str(outdir), "blur", "1", "--", *files], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    for f, a in zip(files, imgs):
        assert_same(read_pgm(str(outdir / os.path.basename(f))), o.blur(a, 1), os.path.basename(f))


def oracle_chain(o, img, chain):
    for verb, args in chain:
        a = [int(x) for x in args if x.lstrip("-").isdigit()]
        if verb == "blur":
            img = o.blur(img, a[0])
        elif verb == "sobel":
            img = o.sobel(img)
        elif verb == "threshold":
            img = o.threshold(img, o.otsu_threshold(img) if args[0] == "otsu" else a[0] & 255)
        elif verb == "adaptive":
            img = o.adaptive_threshold(img, a[0], a[1])
        elif verb == "morph":
            for _ in range(a[0]):
                img = o.dilate(img) if args[0] == "dilate" else o.erode(img)
        elif verb == "resize":
            img = o.resize(img, a[0], a[1])
        elif verb == "crop":
            img = o.crop(img, *a)
    return img


@pytest.mark.gpu
def test_gsbatch_on_gpu_vs_oracle(tmp_path):
    """the real binary (make tool) on MI355X: 12 x 1280x720 + 3 ragged frames through every chain"""
    from oracle.pyoracle import Oracle
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "tool"])
    exe = os.path.join(ROOT, "grayskull_amd", "gsbatch")
    o = Oracle("port")
    files, imgs = [], []
    for k in range(15):
        w, h = (1280, 720) if k < 12 else (317 + k, 203)
        img = Oracle.synth(w, h, 900 + k)
        p = str(tmp_path / ("f%02d.pgm" % k))
        write_pgm(p, img)
        files.append(p)
        imgs.append(img)
    for c, chain in enumerate(CHAINS):
        outdir = tmp_path / ("out%d" % c)
        outdir.mkdir()
        r = subprocess.run([exe, "-v", "-o", str(outdir), *chain_args(chain), "--", *files], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        for f, img in zip(files, imgs):
            got = read_pgm(str(outdir / os.path.basename(f)))
            assert_same(got, oracle_chain(o, img, chain), "chain %d %s" % (c, os.path.basename(f)))


def _random_chain(rs, w, h):
    """a valid random verb chain for a w x h image (tracks the size through resize / crop)"""
    chain = []
    for _ in range(rs.randint(1, 6)):
        v = rs.choice(["blur", "sobel", "threshold", "adaptive", "morph", "resize", "crop"])
        if v == "blur":
            chain.append(("blur", [str(rs.choice([1, 2, 3, 4, 7]))]))
        elif v == "sobel":
            if w >= 3 and h >= 3:
                chain.append(("sobel", []))
        elif v == "threshold":
            chain.append(("threshold", [str(rs.randint(1, 300))]))      # otsu can legitimately fail (t == 0)
        elif v == "adaptive":
            chain.append(("adaptive", [str(rs.randint(1, 9)), str(rs.randint(0, 12))]))
        elif v == "morph":
            chain.append(("morph", [str(rs.choice(["erode", "dilate"])), str(rs.randint(1, 4))]))
        elif v == "resize":
            w, h = int(rs.randint(8, 80)), int(rs.randint(8, 60))
            chain.append(("resize", [str(w), str(h)]))
        elif v == "crop" and w > 8 and h > 8:
            cw, ch = int(rs.randint(4, w)), int(rs.randint(4, h))
            x, y = int(rs.randint(0, w - cw + 1)), int(rs.randint(0, h - ch + 1))
            chain.append(("crop", [str(x), str(y), str(cw), str(ch)]))
            w, h = cw, ch
    return chain or [("blur", ["1"])]


@pytest.mark.skipif(not os.path.exists(NANO), reason="reference checkout not present")
def test_gsbatch_random_chains_equal_piped_nanomagick_emulated(tmp_path):
    """16 random verb chains over three files of two sizes, against the reference CLI run verb by verb"""
    exe, nano = build_emu(tmp_path), build_ref_nano(tmp_path)
    rs = np.random.RandomState(2024)
    files = []
    for k, (w, h) in enumerate([(64, 48), (64, 48), (37, 29)]):
        p = str(tmp_path / ("rnd%d.pgm" % k))
        a = rs.randint(0, 256, (h, w)).astype(np.uint8)
        a[0, 0] = max(int(a[0, 0]), 33)          # keep the reference's reader happy (see the reader test)
        write_pgm(p, a)
        files.append(p)
    for c in range(16):
        chain = _random_chain(rs, 37, 29)        # valid for the smaller size, hence for both
        outdir = tmp_path / ("rc%d" % c)
        outdir.mkdir()
        r = subprocess.run([exe, "-o", str(outdir), *chain_args(chain), "--", *files], capture_output=True, timeout=900)
        assert r.returncode == 0, (chain, r.stderr.decode()[-500:])
        for i, f in enumerate(files):
            exp, err = nano_chain(nano, chain, f, tmp_path, "rcref%d_%d" % (c, i))
            assert exp is not None, (chain, err)
            assert open(str(outdir / os.path.basename(f)), "rb").read() == open(exp, "rb").read(), (chain, f)


@pytest.mark.gpu
def test_gsbatch_feature_verbs_on_gpu(tmp_path):
    """keypoints / faces chains through the real binary on MI355X: records vs the oracle, drawn PGMs vs the
    reference CLI's own build (oracle/_ref/nano_ref, prebuilt in the build container) where it is shipped"""
    from oracle.pyoracle import Oracle
    from grayskull_amd.cascade import Cascade
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "tool"])
    exe = os.path.join(ROOT, "grayskull_amd", "gsbatch")
    nano = os.path.join(ROOT, "oracle", "_ref", "nano_ref")
    o, casc = Oracle("port"), Cascade.from_blob(CASCADE)
    files = [os.path.join(ROOT, "tests", "golden", "lena.pgm")]
    for k, (w, h) in enumerate([(640, 480), (640, 480), (640, 480), (333, 207)]):
        p = str(tmp_path / ("synth%d.pgm" % k))
        a = Oracle.synth(w, h, 60 + k)
        a[0, 0] = max(int(a[0, 0]), 33)
        write_pgm(p, a)
        files.append(p)
    for c, chain in enumerate(FEATURE_CHAINS):
        outdir = tmp_path / ("fout%d" % c)
        outdir.mkdir()
        r = subprocess.run([exe, "-v", "--cascade", CASCADE, "-o", str(outdir), *chain_args(chain), "--", *files],
                           capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        for i, f in enumerate(files):
            check_feature_outputs(outdir, f, chain, o, casc)
            if os.path.exists(nano):
                exp, err = nano_chain(nano, chain, f, tmp_path, "gref%d_%d" % (c, i))
                if exp is not None:
                    assert open(str(outdir / os.path.basename(f)), "rb").read() == open(exp, "rb").read(), (c, f)


@pytest.mark.gpu
def test_gsbatch_orb_verb_on_gpu(tmp_path):
    """`orb <template>` through the real binary on MI355X against the reference CLI's own build (prebuilt)"""
    nano = os.path.join(ROOT, "oracle", "_ref", "nano_ref")
    if not os.path.exists(nano):
        pytest.skip("oracle/_ref/nano_ref was not prebuilt")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "tool"])
    exe = os.path.join(ROOT, "grayskull_amd", "gsbatch")
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    a = read_pgm(lena)
    files = [lena]
    for k, (dx, dy) in enumerate([(5, 3), (1, 9)]):
        b = np.zeros_like(a)
        b[:-dy, :-dx] = a[dy:, dx:]
        b[0, 0] = 40
        p = str(tmp_path / ("shift%d.pgm" % k))
        write_pgm(p, b)
        files.append(p)
    outdir = tmp_path / "orb"
    outdir.mkdir()
    r = subprocess.run([exe, "-o", str(outdir), "orb", lena, "--", *files], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    for i, f in enumerate(files):
        out = str(tmp_path / ("ref%d.pgm" % i))
        rr = subprocess.run([nano, "orb", lena, f, out], capture_output=True, timeout=600)
        assert rr.returncode == 0
        assert open(str(outdir / (os.path.basename(f) + ".orb.txt"))).read().splitlines()[0] + "\n" == rr.stdout.decode()
        assert open(str(outdir / os.path.basename(f)), "rb").read() == open(out, "rb").read(), f


def test_gsbatch_two_workers_asan_clean(tmp_path):
    """the C driver with two worker threads (two emulated devices), every kind of chain, under AddressSanitizer
    with leak detection: no invalid access, nothing left behind when the workers exit (per-thread library
    context with a destructor) -- the host logic of `--gpus N`, which no 1-GPU box can run for real"""
    emu_dir = os.path.join(ROOT, "tests", "emu")
    csrc = os.path.join(ROOT, "grayskull_amd", "csrc")
    lib = str(tmp_path / "libgs_kernel_emu.so")
    san = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.check_call(["g++", "-DGS_EMU", "-DGS_BOXR_MAX=3", *san, "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w",
                           "-I" + emu_dir, "-I" + csrc, *[os.path.join(csrc, u + ".cpp") for u in ("gs_ctx", "gs_stencil", "gs_detect", "gs_comm", "gs_fused", "gs_box")],
                           os.path.join(emu_dir, "hip_emu.cpp"), "-o", lib])
    exe = str(tmp_path / "gsbatch_asan")
    subprocess.check_call(["gcc", "-std=c99", *san, "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L" + str(tmp_path), "-lgs_kernel_emu", "-pthread", "-Wl,-rpath," + str(tmp_path)])
    lena = os.path.join(ROOT, "tests", "golden", "lena.pgm")
    env = dict(os.environ, GS_EMU_DEVICES="2", ASAN_OPTIONS="detect_stack_use_after_return=0:detect_leaks=1:exitcode=66")
    for c, chain in enumerate([["blur", "2", ":", "sobel", ":", "faces", "1"], ["blur", "1", ":", "threshold", "otsu", ":", "morph", "dilate", "2"],
                               ["keypoints", "50", "20"], ["orb", lena]]):
        out = tmp_path / ("asan%d" % c)
        out.mkdir()
        r = subprocess.run([exe, "--gpus", "2", "--cascade", CASCADE, "-o", str(out), *chain, "--", lena, lena, lena],
                           capture_output=True, timeout=900, env=env)
        err = r.stderr.decode()
        assert r.returncode == 0 and "ERROR: AddressSanitizer" not in err and "LeakSanitizer" not in err, err[-1500:]


def _digest_of(sums):
    d = 1469598103934665603
    for v in sums:
        d = ((d ^ int(v)) * 1099511628211) & 0xffffffffffffffff
    return d


def _wsum(a):
    b = np.ascontiguousarray(a).view(np.uint8).reshape(-1).astype(np.uint64)
    return int(np.sum(np.arange(1, b.size + 1, dtype=np.uint64) * (b + np.uint64(1)), dtype=np.uint64))


@pytest.mark.parametrize("gpus", [1, 2, 3, 8])  # 8: the world size north_star is quoted on; three workers own no file
def test_gsbatch_collectives_checksum_of_checksums_emulated(tmp_path, gpus):
    """SURVEY 8(e) in the C driver: the cascade blob is broadcast, every worker's per-file counts and output checksums
    are all-gathered, the wall time all-reduced (gsh_comm_*: RCCL on GPUs, a rendezvous of the emulated devices' threads
    here).  `-v` prints the checksum of checksums over the files in command-line order: the same for 1, 2, 3 and 8
    workers, and equal to the digest bench.py computes (FNV-style fold of the per-frame wsum of the oracle's outputs)"""
    import re
    from oracle.pyoracle import Oracle
    exe = build_emu(tmp_path)
    o = Oracle("port")
    files, sums = [], []
    for k, (w, h) in enumerate([(160, 120), (160, 120), (131, 101), (160, 120), (160, 120)]):
        p = str(tmp_path / ("cs%d.pgm" % k))
        a = Oracle.synth(w, h, 1000 + k)
        write_pgm(p, a)
        files.append(p)
        e = o.sobel(o.blur(a, 2))
        sums.append(_wsum(o.threshold(e, o.otsu_threshold(e))))
    outdir = tmp_path / "cs_out"
    outdir.mkdir()
    env = dict(os.environ, GS_EMU_DEVICES=str(gpus))
    r = subprocess.run([exe, "-v", "--gpus", str(gpus), "-o", str(outdir), "blur", "2", ":", "sobel", ":", "threshold", "otsu", "--", *files],
                       capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    m = re.search(rb"collectives: (.*?) \| checksum of checksums ([0-9a-f]{16}) over (\d+) file\(s\), (\d+) result record", r.stderr)
    assert m, r.stderr.decode()[-800:]
    assert int(m.group(3)) == len(files) and int(m.group(2), 16) == _digest_of(sums), (m.group(2), "%016x" % _digest_of(sums))
    # one worker: local copies; several: the host-rendezvous backend (the one real GPUs fall back to without librccl)
    assert (b"local copies" if gpus == 1 else b"host rendezvous of %d worker threads" % gpus) in m.group(1), m.group(1)
    # faces: the blob reaches every worker through the broadcast; counts come back through the all-gather
    from grayskull_amd.cascade import Cascade
    casc = Cascade.from_blob(CASCADE)
    outdir2 = tmp_path / "cs_faces"
    outdir2.mkdir()
    r = subprocess.run([exe, "-v", "--gpus", str(gpus), "--cascade", CASCADE, "-o", str(outdir2), "faces", "1", "--", *files],
                       capture_output=True, timeout=1800, env=env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    m = re.search(rb"over (\d+) file\(s\), (\d+) result record", r.stderr)
    want = sum(len(o.lbp_detect(casc, o.integral(read_pgm(f)), 100, 1.2, 1.0, 4.0, 1)) for f in files)
    assert m and int(m.group(2)) == want, (m and m.group(2), want)


@pytest.mark.gpu
def test_gsbatch_rccl_world1_on_gpu(tmp_path):
    """the C driver's collectives over REAL RCCL (librccl.so looked up at run time, ncclCommInitAll over one device):
    broadcast of the cascade blob, all-gather of counts + checksums, all-reduce(max) of the wall time.  The digest `-v`
    prints equals the FNV fold of the per-file wsum of the oracle's outputs, i.e. bench.py's output_checksum_of_checksums"""
    import re
    from oracle.pyoracle import Oracle
    from grayskull_amd.cascade import Cascade
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "grayskull_amd", "csrc"), "tool"])
    exe = os.path.join(ROOT, "grayskull_amd", "gsbatch")
    o = Oracle("port")
    files, sums = [], []
    for k in range(6):
        w, h = (640, 360) if k < 4 else (333, 201)
        a = Oracle.synth(w, h, 1000 + k)
        p = str(tmp_path / ("r%d.pgm" % k))
        write_pgm(p, a)
        files.append(p)
        e = o.sobel(o.blur(a, 2))
        sums.append(_wsum(o.threshold(e, o.otsu_threshold(e))))
    out = tmp_path / "o1"
    out.mkdir()
    env = dict(os.environ, GS_COMM_RCCL="1")  # a world of one takes local copies unless asked: here the real library is the point
    r = subprocess.run([exe, "-v", "--gpus", "1", "-o", str(out), "blur", "2", ":", "sobel", ":", "threshold", "otsu", "--", *files],
                       capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    m = re.search(rb"collectives: (.*?) \| checksum of checksums ([0-9a-f]{16}) over (\d+) file", r.stderr)
    assert m, r.stderr.decode()[-800:]
    assert m.group(1).startswith(b"rccl "), m.group(1)           # the real library, not the local fallback
    assert int(m.group(2), 16) == _digest_of(sums)
    casc = Cascade.from_blob(CASCADE)
    out2 = tmp_path / "o2"
    out2.mkdir()
    r = subprocess.run([exe, "-v", "--gpus", "1", "--cascade", CASCADE, "-o", str(out2), "faces", "1", "--", *files[:3]],
                       capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    m = re.search(rb"collectives: (rccl .*?) \| .* over (\d+) file\(s\), (\d+) result record", r.stderr)
    want = sum(len(o.lbp_detect(casc, o.integral(read_pgm(f)), 100, 1.2, 1.0, 4.0, 1)) for f in files[:3])
    assert m and int(m.group(3)) == want, (r.stderr.decode()[-400:], want)
