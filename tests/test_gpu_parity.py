"""Parity tests proper: the HIP path on a real MI355X, called through the C ABI, against the CPU
oracle on the same seeded inputs -- bit-exact for every integer/byte/index output; the ORB angle
is within 1e-5 (and in fact identical bits, same host libm).  Also the reference-generated golden
vectors at BASELINE.json's full sizes and size-independent properties on batches."""
import os
import subprocess

import numpy as np
import pytest

import parity_cases as pc
from oracle.pyoracle import Oracle
from util import assert_same, fnv, lena, random_cascade

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST, DEV = pc.Mem("host"), pc.Mem("device")
SENT = pc.SENTINEL

SHAPES = [(67, 45), (64, 40), (16, 16), (1040, 7), (2064, 5), (33, 3), (1, 1), (3, 3), (16, 1),
          (640, 480), (1280, 720), (1001, 333), (4095, 9), (4097, 5), (4112, 6)]


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", SHAPES)
def test_stencils(hip, oracle, shape, mem):
    w, h = shape
    pc.stencils(hip, oracle, Oracle.synth(w, h, w * 7 + h), mem)
    rs = np.random.RandomState(w + h)
    pc.stencils(hip, oracle, rs.randint(0, 256, (h, w)).astype(np.uint8), mem, radii=(1, 2, 3, 9))


def test_stencils_extremes(hip, oracle):
    for mem in (HOST, DEV):
        for v in (0, 255):
            pc.stencils(hip, oracle, np.full((9, 48), v, np.uint8), mem)
        chk = ((np.indices((10, 32)).sum(0) % 2) * 255).astype(np.uint8)
        pc.stencils(hip, oracle, chk, mem)
        pc.stencils(hip, oracle, Oracle.synth(20, 9, 3), mem, radii=(0, 20, 1000))
        pc.stencils(hip, oracle, Oracle.synth(2048, 64, 4), mem, radii=(1, 2, 3, 20))


def test_strip_kernels_on_unaligned_device_views(hip, oracle):
    """frames at odd byte offsets inside a device buffer must fall back correctly"""
    import torch
    img = Oracle.synth(64, 40, 1)
    for off in (1, 4, 16):
        buf = torch.zeros(64 * 40 * 2 + 64, dtype=torch.uint8, device="cuda")
        s = buf[off:off + 64 * 40].view(40, 64)
        s.copy_(torch.from_numpy(img))
        d = torch.full((64 * 40 + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        dv = d[off:off + 64 * 40].view(40, 64)
        hip.blur(dv, s, 2)
        assert_same(dv.cpu().numpy(), oracle.blur(img, 2), "blur at offset %d" % off)
        assert int(d[off - 1]) == 0xAB and int(d[off + 64 * 40]) == 0xAB
        hip.threshold(dv, 90)
        assert_same(dv.cpu().numpy(), oracle.threshold(oracle.blur(img, 2), 90), "threshold at offset %d" % off)
        assert int(d[off - 1]) == 0xAB and int(d[off + 64 * 40]) == 0xAB
        assert_same(hip.histogram(s), oracle.histogram(img), "hist at offset %d" % off)
        # every kernel with an alignment-dependent fast path must take its fallback here
        hip.sobel(dv, s)
        assert_same(dv.cpu().numpy()[1:-1, 1:-1], oracle.sobel(img)[1:-1, 1:-1], "sobel at offset %d" % off)
        for name in ("erode", "dilate"):
            getattr(hip, name)(dv, s)
            assert_same(dv.cpu().numpy(), getattr(oracle, name)(img), "%s at offset %d" % (name, off))
        hip.blur(dv, s, 9)
        assert_same(dv.cpu().numpy(), oracle.blur(img, 9), "blur r=9 at offset %d" % off)
        hip.adaptive_threshold(dv, s, 5, 3)
        assert_same(dv.cpu().numpy(), oracle.adaptive_threshold(img, 5, 3), "adaptive at offset %d" % off)
        k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.int8)
        hip.filter(dv, s, k, 16)
        assert_same(dv.cpu().numpy(), oracle.filter(img, k, 16), "filter at offset %d" % off)
        assert int(d[off - 1]) == 0xAB and int(d[off + 64 * 40]) == 0xAB
        half = torch.full((32 * 20 + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        hv = half[off:off + 32 * 20].view(20, 32)
        hip.downsample(hv, s)
        assert_same(hv.cpu().numpy(), oracle.downsample(img), "downsample at offset %d" % off)
        assert int(half[off - 1]) == 0xAB and int(half[off + 32 * 20]) == 0xAB
        iib = torch.zeros(64 * 40 * 4 + 64, dtype=torch.uint8, device="cuda")
        iiv = iib[4 * off:4 * off + 64 * 40 * 4].view(torch.int32).view(40, 64)
        hip.integral(s, iiv)
        assert_same(iiv.cpu().numpy().view(np.uint32), oracle.integral(img), "integral at offset %d" % off)
        t = img[5:13, 7:19].copy()
        res = torch.zeros((40 - 8 + 1, 64 - 12 + 1), dtype=torch.uint8, device="cuda")
        hip.match_template(s, torch.from_numpy(t).cuda(), res)
        assert_same(res.cpu().numpy(), oracle.match_template(img, t), "match_template at offset %d" % off)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", [(67, 45), (64, 40), (1, 1), (5, 1), (1031, 3), (1920, 1080), (4096, 512)])
def test_pointwise_and_integral(hip, oracle, shape, mem):
    w, h = shape
    img = Oracle.synth(w, h, 11 + w)
    pc.pointwise(hip, oracle, img, mem)
    pc.integral(hip, oracle, img, mem)
    pc.integral(hip, oracle, np.full((h, w), 255, np.uint8), mem)


def test_otsu_scan_float_order(hip, oracle):
    rs = np.random.RandomState(5)
    for _ in range(8):
        img = rs.choice(256, size=(2160, 3840), p=rs.dirichlet(np.ones(256) * 0.3)).astype(np.uint8)
        assert hip.otsu_threshold(img) == oracle.otsu_threshold(img)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", [(67, 45), (640, 480)])
def test_next_rows(hip, oracle, shape, mem):
    w, h = shape
    pc.next_rows(hip, oracle, Oracle.synth(w, h, 21), mem)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", [(67, 45), (96, 80), (7, 7), (6, 30), (40, 8), (640, 480), (1283, 517), (260, 17), (8, 8), (1284, 100)])
def test_fast(hip, oracle, shape, mem):
    w, h = shape
    for strip in (0, 2):  # gsh_tune key 7: 0 LDS-tile score kernel (default), 2 global byte loads
        hip.tune(7, strip)
        try:
            pc.fast(hip, oracle, Oracle.synth(w, h, 5), mem)
            rs = np.random.RandomState(1)
            pc.fast(hip, oracle, rs.randint(0, 256, (h, w)).astype(np.uint8), mem, threshold=5, caps=(5000, 1))
            pc.fast(hip, oracle, rs.randint(0, 40, (h, w)).astype(np.uint8), mem, threshold=30)
        finally:
            hip.tune(7, 0)


def test_fast_tile_kernel_equals_per_pixel_kernel(hip, oracle):
    """k_fast_score_q4 (default) and k_fast_score_px (gsh_tune key 7 = 2) against the oracle on a 1280x720 frame with a dark
    (p < t) region and random texture, thresholds incl. one that puts every pixel in the wrap class"""
    rs = np.random.RandomState(11)
    img = Oracle.synth(1280, 720, 9)
    img[80:200, 100:400] = rs.randint(0, 12, (120, 300))
    img[500:620, 1100:1280] = rs.randint(0, 256, (120, 180))
    for t in (20, 3, 200, 300):
        for force_px in (0, 2):
            hip.tune(7, force_px)
            try:
                pc.fast(hip, oracle, img, DEV, threshold=t, caps=(5000,))
            finally:
                hip.tune(7, 0)


def test_fast_both_nms_kernels(hip, oracle):
    """pass 2: the sparse kernel behind the score kernel's bitmap (default) and the item-by-item kernel (key 19 = 1): the
    oracle's keypoints in the oracle's order, caps that cut the list, a caller's non-zero score-map frame, widths that are
    no multiple of 64 / 16"""
    rs = np.random.RandomState(5)
    for (w, h) in ((1280, 720), (1000, 333), (70, 71), (131, 64)):
        flat = np.full((h, w), 100, np.uint8)
        flat[::3, ::3] = 140
        for img in (Oracle.synth(w, h, 8), rs.randint(0, 256, (h, w)).astype(np.uint8), flat):
            for key19 in (0, 1):
                hip.tune(19, key19)
                try:
                    pc.fast(hip, oracle, img, DEV, threshold=12, caps=(30000, 9, 1))
                finally:
                    hip.tune(19, 0)


def test_fast_quirk(hip, oracle):
    pc.fast_unsigned_wrap_quirk(hip, oracle, HOST)
    pc.fast_unsigned_wrap_quirk(hip, oracle, DEV)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", [(96, 80), (67, 45), (640, 480)])
def test_orb_and_match(hip, oracle, shape, mem):
    w, h = shape
    pc.orb(hip, oracle, Oracle.synth(w, h, 7), mem, nkps=50 if w < 600 else 500)


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
def test_lbp(hip, oracle, cascade, mem):
    pc.lbp(hip, oracle, Oracle.synth(96, 80, 7), mem, cascade,
           params=((4096, 1.1, 1.0, 4.0, 1), (10, 1.3, 1.0, 2.0, 3)),
           windows=((0, 0, 1.0), (5, 3, 1.2), (72, 56, 1.0), (73, 56, 1.0), (0, 0, 3.4)))
    rc = random_cascade(1)
    pc.lbp(hip, oracle, Oracle.synth(320, 200, 9), mem, rc,
           params=((4096, 1.25, 1.0, 2.0, 2), (37, 1.25, 1.0, 2.0, 1), (1, 1.5, 1.0, 1.6, 1), (100000, 1.2, 1.0, 3.0, 1)),
           windows=((0, 0, 1.0), (1, 0, 1.0), (0, 1, 1.5), (40, 24, 1.0)))
    pc.lbp(hip, oracle, Oracle.synth(640, 480, 3), mem, cascade, params=((4096, 1.2, 1.0, 4.0, 2),))
    # config-5 style input (sobel edge map): thousands of hits, max_rects reached in early scales
    edges = oracle.sobel(oracle.blur(Oracle.synth(960, 540, 1000), 2))
    pc.lbp(hip, oracle, edges, mem, cascade, params=((4096, 1.1, 1.0, 4.0, 1), (100, 1.1, 1.0, 4.0, 1), (100000, 1.3, 1.0, 3.0, 3)))


# ---- golden vectors generated by the unmodified reference, at the BASELINE.json sizes --------
def test_kat_lena(hip, kat):
    k, img = kat["lena"], lena()
    a, b = np.zeros_like(img), np.zeros_like(img)
    for r, hsh in k["blur"].items():
        hip.blur(a, img, int(r))
        assert fnv(a) == hsh
    hip.blur(a, img, 2)
    hip.sobel(b, a)
    assert fnv(b) == k["blur_sobel"]
    assert hip.otsu_threshold(img) == k["otsu_src"]
    c = img.copy()
    hip.threshold(c, k["otsu_src"])
    assert fnv(c) == k["thr_src"]
    for name in ("sobel", "erode", "dilate"):
        d = np.zeros_like(img)
        getattr(hip, name)(d, img)
        assert fnv(d) == k[name]
    ii = hip.integral(img)
    assert fnv(ii) == k["integral"] and int(ii[-1, -1]) == k["integral_last"]
    d = np.zeros_like(img)
    hip.adaptive_threshold(d, img, 15, 5)
    assert fnv(d) == k["adaptive_r15_c5"]
    # BASELINE configs[0]: lena -> gs_resize 512x512 (float32 bilinear) -> gs_blur(2) -> gs_sobel into zeros
    big = np.zeros((512, 512), np.uint8)
    hip.resize(big, img)
    a5, b5 = np.zeros_like(big), np.zeros_like(big)
    hip.blur(a5, big, 2)
    hip.sobel(b5, a5)
    assert fnv(b5) == k["resize512_blur2_sobel"]


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4])
def test_kat_synth_config2_chain(hip, kat, idx):
    """config 2: gs_blur(r=2) -> gs_sobel (zeroed dst) -> gs_otsu_threshold -> gs_threshold"""
    import torch
    k = kat["synth"][idx]
    w, h = k["w"], k["h"]
    src = torch.zeros((1, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(src, k["seed"])
    img = src[0].cpu().numpy()
    assert fnv(img) == k["src"], "device generator == CPU generator"
    a = torch.zeros_like(src[0])
    b = torch.zeros_like(src[0])
    hip.blur(a, src[0], 2)
    assert fnv(a.cpu().numpy()) == k["blur2"]
    hip.sobel(b, a)
    assert fnv(b.cpu().numpy()) == k["blur_sobel"]
    t = hip.otsu_threshold(b)
    assert t == k["otsu"]
    hip.threshold(b, t)
    assert fnv(b.cpu().numpy()) == k["thr"]
    for name in ("sobel", "erode", "dilate"):
        d = torch.zeros_like(src[0])
        getattr(hip, name)(d, src[0])
        assert fnv(d.cpu().numpy()) == k[name], name
    ii = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    hip.integral(src[0], ii)
    assert fnv(ii.cpu().numpy()) == k["integral"]
    # the fused batch entry point must give the same bytes
    tmp, out = torch.zeros_like(src), torch.full_like(src, 9)
    hist = torch.zeros((1, 256), dtype=torch.int32, device="cuda")
    thr = torch.zeros(1, dtype=torch.uint8, device="cuda")
    hip.edge_pipeline_batch(out, tmp, src, 2, hist, thr)
    assert int(thr[0]) == k["otsu"] and fnv(out[0].cpu().numpy()) == k["thr"]
    assert fnv(tmp[0].cpu().numpy()) == k["blur2"]
    # ... and so must the single fused kernel (tmp=NULL: the blurred image stays in registers)
    out2, thr2 = torch.full_like(src, 5), torch.zeros(1, dtype=torch.uint8, device="cuda")
    hist2 = torch.zeros((1, 256), dtype=torch.int32, device="cuda")
    hip.edge_pipeline_batch(out2, None, src, 2, hist2, thr2)
    assert int(thr2[0]) == k["otsu"] and fnv(out2[0].cpu().numpy()) == k["thr"]
    assert bool((hist2 == hist).all()), "fused histogram == histogram of the sobel image"


@pytest.mark.parametrize("idx", [2, 3, 4])
def test_kat_synth_features(hip, kat, cascade, idx):
    """configs 3 and 4: integral + LBP cascade, FAST, ORB on the reference-generated vectors"""
    k = kat["synth"][idx]
    img = Oracle.synth(k["w"], k["h"], k["seed"])
    p = kat["fast_params"]
    sm = np.zeros_like(img)
    kps = hip.fast(img, sm, p["nkps"], p["threshold"])
    assert len(kps) == k["fast"]["n"] and fnv(kps) == k["fast"]["kp"] and fnv(sm) == k["fast"]["scoremap"]
    p = kat["orb_params"]
    ok = hip.orb_extract(img, p["nkps"], p["threshold"], np.zeros_like(img))
    assert len(ok) == k["orb"]["n"] and fnv(ok) == k["orb"]["kp"]
    p = kat["lbp_params"]
    r = hip.lbp_detect(cascade, hip.integral(img), p["max_rects"], p["scale_factor"], p["min_scale"],
                       p["max_scale"], p["step"])
    assert len(r) == k["lbp"]["n"] and fnv(r) == k["lbp"]["rects"]


def test_kat_orb_match(hip, kat):
    k = kat["orb_match"]
    A = Oracle.synth(k["w"], k["h"], k["seed"])
    B = np.zeros_like(A)
    sx, sy = k["shift"]
    B[:k["h"] - sy, :k["w"] - sx] = A[sy:, sx:]
    ka = hip.orb_extract(A, k["nkps"], k["threshold"], np.zeros_like(A))
    kb = hip.orb_extract(B, k["nkps"], k["threshold"], np.zeros_like(A))
    m = hip.match_orb(ka, kb, k["max_matches"], k["max_distance"])
    assert len(m) == k["n"] and fnv(m) == k["matches"]
    ka = hip.orb_extract(A, 2500, k["threshold"], np.zeros_like(A))
    kb = hip.orb_extract(B, 2500, k["threshold"], np.zeros_like(A))
    assert len(hip.match_orb(ka, kb, 2500, k["max_distance"])) == k["n_nkps2500"]


# ---- batches at full size: oracle on a sample of frames + size-independent properties ---------
def test_batch_4k_properties(hip, oracle):
    import torch
    n, h, w = 12, 2160, 3840
    src = torch.zeros((n, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(src, 1000)
    assert np.array_equal(src[5].cpu().numpy(), Oracle.synth(w, h, 1005))
    blur, sob, ero, dil = (torch.zeros_like(src) for _ in range(4))
    hip.blur_batch(blur, src, 2)
    hip.sobel_batch(sob, blur)
    hip.erode_batch(ero, src)
    hip.dilate_batch(dil, src)
    hip.sync()
    for f in (0, 7, 11):  # oracle on a sample
        s = src[f].cpu().numpy()
        b = oracle.blur(s, 2)
        assert_same(blur[f].cpu().numpy(), b, "blur_batch frame %d" % f)
        assert_same(sob[f].cpu().numpy(), oracle.sobel(b), "sobel_batch frame %d" % f)
        assert_same(ero[f].cpu().numpy(), oracle.erode(s), "erode_batch frame %d" % f)
        assert_same(dil[f].cpu().numpy(), oracle.dilate(s), "dilate_batch frame %d" % f)
    # properties over the whole batch
    assert bool((ero <= src).all()) and bool((src <= dil).all())
    assert int(sob[:, 0, :].max()) == 0 and int(sob[:, :, 0].max()) == 0  # frame never written
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    hip.histogram_batch(src, hist)
    assert bool((hist.sum(1) == w * h).all())
    # checksum of checksums: the per-frame device checksums equal the CPU ones for every frame
    sums = torch.zeros(n, dtype=torch.int64, device="cuda")
    hip.checksum_batch(blur, sums)
    idx = np.arange(1, w * h + 1, dtype=np.uint64)
    for f in (0, 3):
        exp = np.sum(idx * (oracle.blur(src[f].cpu().numpy(), 2).reshape(-1).astype(np.uint64) + 1), dtype=np.uint64)
        assert np.uint64(sums[f].cpu().numpy().view(np.uint64)) == exp
    # threshold is idempotent, and edge_pipeline == the separate calls
    thr = torch.zeros(n, dtype=torch.uint8, device="cuda")
    out, tmp = torch.full_like(src, 3), torch.zeros_like(src)
    hip.edge_pipeline_batch(out, tmp, src, 2, hist, thr)
    once = out.clone()
    hip.threshold_batch(out, thr)
    assert bool((once == out).all())
    assert bool(((out == 0) | (out == 255)).all())
    for f in (2, 9):
        s = oracle.sobel(oracle.blur(src[f].cpu().numpy(), 2))
        t = oracle.otsu_threshold(s)
        assert int(thr[f]) == t
        assert_same(once[f].cpu().numpy(), oracle.threshold(s, t), "edge pipeline frame %d" % f)
    # fused kernel path over the whole batch, every radius it supports, == the unfused path
    for r in (1, 2, 3):
        a, b = torch.full_like(src, 1), torch.full_like(src, 2)
        ha, hb = torch.zeros_like(hist), torch.zeros_like(hist)
        ta, tb = torch.zeros_like(thr), torch.zeros_like(thr)
        hip.edge_pipeline_batch(a, tmp, src, r, ha, ta)
        hip.edge_pipeline_batch(b, None, src, r, hb, tb)
        assert bool((a == b).all()) and bool((ha == hb).all()) and bool((ta == tb).all()), "fused r=%d" % r


@pytest.mark.parametrize("shape,levels,nkps", [((1280, 720), 3, 500), ((640, 480), 4, 300), ((130, 70), 4, 50)])
def test_orb_pyramid_device_resident(hip, oracle, shape, levels, nkps):
    """the reference CLI's pyramid ORB driver (nanomagick.c:245-290) with all levels on the device"""
    w, h = shape
    pc.orb_pyramid(hip, oracle, Oracle.synth(w, h, 21), pc.Mem("device"), nkps=nkps, levels=levels)


def test_blur_sobel_batch(hip, oracle, kat):
    """gsh_blur_sobel_batch on the 4K KAT frame: the reference-generated blur2->sobel hash"""
    import torch
    k = [e for e in kat["synth"] if e["w"] == 3840][0]
    src = torch.from_numpy(np.stack([Oracle.synth(3840, 2160, k["seed"]), Oracle.synth(3840, 2160, 9)])).cuda()
    dst = torch.full_like(src, 3)
    hip.blur_sobel_batch(dst, src, 2)
    assert fnv(dst[0].cpu().numpy()) == k["blur_sobel"]
    for r in (1, 3, 4):
        small = src[:, :300, :640].contiguous()
        d = torch.full_like(small, 5)
        hip.blur_sobel_batch(d, small, r)
        assert_same(d[1].cpu().numpy(), oracle.sobel(oracle.blur(small[1].cpu().numpy(), r)), "blur_sobel r=%d" % r)


def test_box_and_filter_kernels_at_frame_sizes(hip, oracle):
    """k_box16 (sliding box sums) and k_filter16 on real frame sizes: 720p against the oracle, and on a
    4K batch the sliding route against the independent integral-image route (gsh_tune key 6 = 3)"""
    import torch
    img = Oracle.synth(1280, 720, 4)
    src = torch.from_numpy(img).cuda()
    d = torch.zeros_like(src)
    for r in (4, 9, 40):
        hip.blur(d, src, r)
        assert_same(d.cpu().numpy(), oracle.blur(img, r), "gs_blur r=%d 720p" % r)
    hip.adaptive_threshold(d, src, 15, 5)
    assert_same(d.cpu().numpy(), oracle.adaptive_threshold(img, 15, 5), "gs_adaptive_threshold r=15 720p")
    for k, norm in (([[1, 2, 1], [2, 4, 2], [1, 2, 1]], 16), ([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], 1), ([[-2, -1, 0], [-1, 1, 1], [0, 1, 2]], 3)):
        k = np.array(k, np.int8)
        hip.filter(d, src, k, norm)
        assert_same(d.cpu().numpy(), oracle.filter(img, k, norm), "gs_filter 720p norm=%d" % norm)
    big = torch.empty((3, 2160, 3840), dtype=torch.uint8, device="cuda")
    hip.synth_batch(big, 77)
    a, b = torch.zeros_like(big), torch.zeros_like(big)
    try:
        for r in (4, 15, 31, 32, 56, 100, 127):  # 31 / 32: last radius of the one-multiply quotient / first of the float one
            hip.tune(6, 0)
            hip.blur_batch(a, big, r)
            hip.tune(6, 3)
            hip.blur_batch(b, big, r)
            assert bool((a == b).all()), "blur r=%d: sliding vs integral route" % r
            hip.tune(6, 0)
            hip.adaptive_threshold_batch(a, big, r, 7)
            hip.tune(6, 3)
            hip.adaptive_threshold_batch(b, big, r, 7)
            assert bool((a == b).all()), "adaptive r=%d: sliding vs integral route" % r
    finally:
        hip.tune(6, 0)


def test_two_host_threads_share_the_library(hip, oracle):
    """SURVEY 8(b) threading: thread-safe per calling thread (thread-local context + stream).  Two
    host threads run different call chains at the same time on their own images."""
    import threading
    import torch
    imgs = [Oracle.synth(640, 480, 300 + i) for i in range(2)]
    out, err = [None, None], []

    def work(i):
        try:
            src = torch.from_numpy(imgs[i]).cuda()
            a, b = torch.zeros_like(src), torch.zeros_like(src)
            for _ in range(20):
                if i == 0:
                    hip.blur(a, src, 2)
                    hip.sobel(b, a)
                    t = hip.otsu_threshold(b)
                    hip.threshold(b, t)
                else:
                    hip.erode(a, src)
                    hip.dilate(b, a)
                    ii = torch.zeros((480, 640), dtype=torch.int32, device="cuda")
                    hip.integral(b, ii)
            hip.sync()
            out[i] = (b.cpu().numpy(), None if i == 0 else ii.cpu().numpy().view(np.uint32))
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    s = oracle.sobel(oracle.blur(imgs[0], 2))
    assert_same(out[0][0], oracle.threshold(s, oracle.otsu_threshold(s)), "thread 0 chain")
    d = oracle.dilate(oracle.erode(imgs[1]))
    assert_same(out[1][0], d, "thread 1 chain")
    assert_same(out[1][1], oracle.integral(d), "thread 1 integral")


@pytest.mark.parametrize("mem", [HOST, DEV], ids=["host", "device"])
@pytest.mark.parametrize("shape", [(640, 480), (67, 45), (131, 20)])
def test_geometry_and_template_matching(hip, oracle, shape, mem):
    w, h = shape
    pc.geometry(hip, oracle, Oracle.synth(w, h, w + h), mem)


def test_template_matching_large(hip, oracle):
    """a 64x64 template cut out of a 1280x720 frame: the maximum sits where it was cut"""
    img = Oracle.synth(1280, 720, 4)
    t = img[200:264, 500:564].copy()
    import torch
    r = torch.zeros((720 - 63, 1280 - 63), dtype=torch.uint8, device="cuda")
    hip.match_template(torch.from_numpy(img).cuda(), torch.from_numpy(t).cuda(), r)
    rn = r.cpu().numpy()
    assert rn[200, 500] == 255, "zero SSD where the template was cut"
    first = int(np.argmax(rn))  # numpy returns the FIRST maximum, like the reference's strict '>'
    assert hip.find_best_match(r) == (first % rn.shape[1], first // rn.shape[1])
    rows = oracle.match_template(img[180:300], t)  # oracle on a 120-row slab (the full frame takes a minute)
    assert_same(rn[180:180 + rows.shape[0]], rows, "template rows 180..")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(300, 200, 128, 128), (400, 260, 64, 64), (517, 131, 33, 17), (1280, 720, 181, 181),
                                  (640, 96, 257, 32), (200, 180, 16, 32), (3840, 400, 256, 64)])
def test_match_template_matrix_core_kernel_on_the_gpu(hip, oracle, case):
    """k_match_template_mfma on real matrix cores (the emulator only models v_mfma_i32_32x32x32_i8): both tilings
    (gsh_tune key 20 = 2 / 3) and the launcher's own choice against the dot-product kernels (key 20 = 1) byte for byte,
    and a slab of rows against the oracle; random, all-255-on-all-0 (largest sums) and cut-out templates"""
    import torch
    iw, ih, tw, th = case
    rs = np.random.RandomState(iw + tw)
    img = rs.randint(0, 256, (ih, iw)).astype(np.uint8)
    pairs = [(img, rs.randint(0, 256, (th, tw)).astype(np.uint8)), (img, img[ih - th:, iw - tw:].copy()),
             (np.zeros_like(img), np.full((th, tw), 255, np.uint8))]
    try:
        for k, (im, t) in enumerate(pairs):
            d_im, d_t = torch.from_numpy(im).cuda(), torch.from_numpy(t).cuda()
            outs = {}
            for key in (1, 0, 2, 3, 8, 4, 5):
                hip.tune(20, key)
                r = torch.zeros((ih - th + 1, iw - tw + 1), dtype=torch.uint8, device="cuda")
                hip.match_template(d_im, d_t, r)
                outs[key] = r.cpu().numpy()
            for key in (0, 2, 3, 8, 4, 5):
                assert_same(outs[key], outs[1], "mfma (key 20 = %d) vs dot4, pair %d, %dx%d on %dx%d" % (key, k, tw, th, iw, ih))
            rows = min(6, ih - th + 1)
            slab = oracle.match_template(im[: th + rows - 1], t)
            assert_same(outs[0][:rows], slab, "mfma vs oracle, first rows, pair %d" % k)
    finally:
        hip.tune(20, 0)


def test_orb_extract_batch(hip, oracle):
    frames = np.stack([Oracle.synth(1280, 720, 4 + i) for i in range(5)])
    frames[2] = 0  # a frame without corners
    pc.orb_batch(hip, oracle, frames, DEV, nkps=500)


def test_pipeline_chunk_overlap(hip, oracle):
    """gsh_edge_pipeline_batch cuts big batches into chunks whose threshold pass runs on a side
    stream under the next chunk's fused kernel: same bytes for every chunking, ragged last chunk,
    repeated calls (side stream rejoined each time), also on a caller-provided stream."""
    import torch
    n, h, w = 23, 96, 256
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(src, 4242)
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    ref, tref = torch.zeros_like(src), torch.zeros(n, dtype=torch.uint8, device="cuda")
    try:
        hip.tune(5, -1)  # never split
        hip.edge_pipeline_batch(ref, None, src, 2, hist, tref)
        hip.sync()
        href = hist.clone()
        for f in (0, 7, 8, n - 1):
            s = oracle.sobel(oracle.blur(src[f].cpu().numpy(), 2))
            t = oracle.otsu_threshold(s)
            assert int(tref[f]) == t
            assert_same(ref[f].cpu().numpy(), oracle.threshold(s, t), "unsplit pipeline frame %d" % f)
        for per in (1, 4, 8, 22, 0):
            hip.tune(5, per)
            for rep in range(2):
                out, thr = torch.full_like(src, 7), torch.zeros(n, dtype=torch.uint8, device="cuda")
                hist.zero_()
                hip.edge_pipeline_batch(out, None, src, 2, hist, thr)
                hip.sync()
                assert bool((out == ref).all()) and bool((thr == tref).all()) and bool((hist == href).all()), \
                    "chunk size %d, call %d" % (per, rep)
        hip.tune(5, 4)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            hip.use_torch_stream()
            out, thr = torch.full_like(src, 3), torch.zeros(n, dtype=torch.uint8, device="cuda")
            hip.edge_pipeline_batch(out, None, src, 2, hist, thr)
            after = out.clone()  # stream-ordered consumer on the caller's stream, no host sync
        st.synchronize()
        assert bool((after == ref).all()) and bool((thr == tref).all())
    finally:
        hip.tune(5, 0)
        hip.use_torch_stream()


def test_pipeline_chunk_ordering_big_batch_and_pinned_host_buffers(hip):
    """the side stream is ordered against the caller's stream by events alone: every frame of a
    multi-chunk batch equals the unsplit path -- on a batch big enough that chunks really overlap (128 x 1080p,
    32-frame chunks) and with src / dst / thr in page-locked HOST memory (host-coherent buffers, where a missing
    release would show first)"""
    import ctypes as C
    import torch
    n, h, w = 128, 1080, 1920
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(src, 777)
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    ref, tref = torch.zeros_like(src), torch.zeros(n, dtype=torch.uint8, device="cuda")
    try:
        hip.tune(5, -1)
        hip.edge_pipeline_batch(ref, None, src, 2, hist, tref)
        hip.sync()
        hip.tune(5, 0)
        for rep in range(3):
            out, thr = torch.full_like(src, 9), torch.zeros(n, dtype=torch.uint8, device="cuda")
            hip.edge_pipeline_batch(out, None, src, 2, hist, thr)
            hip.sync()
            bad = (out != ref).flatten(1).any(1).nonzero().flatten().tolist()
            assert not bad and bool((thr == tref).all()), "32-frame chunks, call %d: frames %s differ" % (rep, bad[:8])
        # page-locked host buffers, small frames, 8-frame chunks
        m, hh, ww = 40, 96, 256
        nb = m * hh * ww
        hsrc, hdst, hthr = hip.c.gsh_host_alloc(nb), hip.c.gsh_host_alloc(nb), hip.c.gsh_host_alloc(m)
        try:
            a_src = np.ctypeslib.as_array(C.cast(hsrc, C.POINTER(C.c_uint8)), (m, hh, ww))
            a_dst = np.ctypeslib.as_array(C.cast(hdst, C.POINTER(C.c_uint8)), (m, hh, ww))
            a_thr = np.ctypeslib.as_array(C.cast(hthr, C.POINTER(C.c_uint8)), (m,))
            small = src[:m, :hh, :ww].contiguous()
            a_src[:] = small.cpu().numpy()
            hs = torch.zeros((m, 256), dtype=torch.int32, device="cuda")
            r2, t2 = torch.zeros_like(small), torch.zeros(m, dtype=torch.uint8, device="cuda")
            hip.tune(5, -1)
            hip.edge_pipeline_batch(r2, None, small, 2, hs, t2)
            hip.sync()
            for per in (8, 3):
                hip.tune(5, per)
                a_dst[:] = 5
                a_thr[:] = 0
                hip.c.gsh_edge_pipeline_batch(hdst, None, hsrc, ww, hh, m, 2, hs.data_ptr(), hthr)
                hip.sync()
                assert np.array_equal(a_dst, r2.cpu().numpy()) and np.array_equal(a_thr, t2.cpu().numpy()), \
                    "pinned host buffers, %d-frame chunks" % per
        finally:
            hip.c.gsh_host_free(hsrc), hip.c.gsh_host_free(hdst), hip.c.gsh_host_free(hthr)
    finally:
        hip.tune(5, 0)


def test_batch_integral_lbp_fast(hip, oracle, cascade):
    import torch
    n, h, w = 3, 480, 640
    frames = np.stack([Oracle.synth(w, h, 50 + i) for i in range(n)])
    src = torch.from_numpy(frames).cuda()
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda")
    hip.integral_batch(src, ii)
    rc = random_cascade(2)
    for casc, cap in ((cascade, 4096), (rc, 300)):
        dc = hip.cascade_create(casc)
        rects = torch.zeros((n, cap, 4), dtype=torch.int32, device="cuda")
        counts = torch.zeros(n, dtype=torch.int32, device="cuda")
        hip.lbp_detect_batch(dc, ii, rects, counts, cap, 1.2, 1.0, 3.0, 2)
        hip.sync()
        for f in range(n):
            ro = oracle.lbp_detect(casc, oracle.integral(frames[f]), cap, 1.2, 1.0, 3.0, 2)
            assert int(counts[f]) == len(ro)
            got = rects[f, :len(ro)].cpu().numpy().view(np.uint32)
            assert_same(got, np.stack([ro["x"], ro["y"], ro["w"], ro["h"]], 1) if len(ro) else got, "lbp batch frame %d" % f)
        dc.close()
    sm = torch.zeros_like(src)
    kps = torch.zeros((n, 800, 12), dtype=torch.int32, device="cuda")
    counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    hip.fast_batch(src, sm, kps, counts, 800, 20)
    hip.sync()
    for f in range(n):
        ko, smo = oracle.fast(frames[f], 800, 20)
        assert int(counts[f]) == len(ko)
        assert_same(kps[f, :len(ko)].cpu().numpy().reshape(-1).view(ko.dtype), ko, "fast batch frame %d" % f)
        assert_same(sm[f].cpu().numpy(), smo, "fast batch scoremap %d" % f)


def test_c99_dropin_program_on_gpu(tmp_path):
    """tests/c/test_dropin.c (strict C99) linked against the real libgrayskull_hip.so"""
    exe = tmp_path / "dropin"
    libdir = os.path.join(ROOT, "grayskull_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "test_dropin.c"),
                           "-o", str(exe), "-L", libdir, "-l:libgrayskull_hip.so", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)]).decode()
    assert "all passed" in out


# ---- limits and odd call patterns ---------------------------------------------------------------
def test_large_frame_8192(hip, oracle):
    """one 8192x8192 frame (67 MB): strip kernels with 8 column blocks, tall bands"""
    import torch
    img = Oracle.synth(8192, 8192, 77)
    s = torch.from_numpy(img).cuda()
    d = torch.full_like(s, SENT)
    hip.sobel(d, s)
    assert_same(d.cpu().numpy(), oracle.sobel(img, np.full_like(img, SENT)), "sobel 8192^2")
    hip.blur(d, s, 2)
    assert_same(d.cpu().numpy(), oracle.blur(img, 2), "blur 8192^2")
    hip.erode(d, s)
    assert_same(d.cpu().numpy(), oracle.erode(img), "erode 8192^2")
    assert_same(hip.histogram(s), oracle.histogram(img), "histogram 8192^2")
    assert hip.otsu_threshold(s) == oracle.otsu_threshold(img)


def test_integral_wraps_mod_2_32_like_the_reference(hip, oracle):
    """255 * 4200 * 4200 > 2^32: the reference's `unsigned` table wraps (grayskull.h:744-752)"""
    img = np.full((4200, 4200), 255, np.uint8)
    ii = hip.integral(img)
    exp = oracle.integral(img)
    assert_same(ii, exp, "integral wrap")
    assert int(exp[-1, -1]) == (255 * 4200 * 4200) % 2 ** 32


@pytest.mark.parametrize("shape", [(3840, 2160, 33), (3838, 1080, 67), (4100, 600, 110), (1282, 720, 300)])
def test_integral_batches_beyond_the_infinity_cache(hip, oracle, shape):
    """source planes of more than 256 MB: gs_integral's first pass reads them with streaming loads (k_integral_colsum<.., NT>),
    and so does the third for rows of more than 2048 px (k_integral_wave<16, .., NT>: whole strips, ragged rows, column chunks
    beyond 4096 px); tune key 6 = 8 takes the default policy -- the tables must be the same either way"""
    import torch
    w, h, n = shape
    assert w * h * n > 256 << 20
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(src, 300)
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda")
    hip.integral_batch(src, ii)
    hip.sync()
    for f in (0, n // 2, n - 1):
        assert_same(ii[f].cpu().numpy().view(np.uint32), oracle.integral(src[f].cpu().numpy()), "integral frame %d of %s" % (f, shape))
    ii2 = torch.zeros_like(ii)
    try:
        hip.tune(6, 8)
        hip.integral_batch(src, ii2)
        hip.sync()
    finally:
        hip.tune(6, 0)
    assert torch.equal(ii, ii2)


def test_more_frames_than_grid_z(hip, oracle):
    """40000 tiny frames in one batch call: launches are split at the grid.z limit"""
    import torch
    n, h, w = 40000, 16, 32
    rs = np.random.RandomState(3)
    frames = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    src = torch.from_numpy(frames).cuda()
    dst = torch.zeros_like(src)
    hip.sobel_batch(dst, src)
    got = dst.cpu().numpy()
    for f in (0, 32767, 32768, 39999):
        assert_same(got[f], oracle.sobel(frames[f]), "sobel frame %d" % f)
    hip.blur_batch(dst, src, 1)
    got = dst.cpu().numpy()
    for f in (0, 32767, 32768, 39999):
        assert_same(got[f], oracle.blur(frames[f], 1), "blur frame %d" % f)
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    thr = torch.zeros(n, dtype=torch.uint8, device="cuda")
    hip.otsu_batch(src, hist, thr)
    t = thr.cpu().numpy()
    for f in (0, 32767, 32768, 39999):
        assert int(t[f]) == oracle.otsu_threshold(frames[f])


def test_mixed_host_and_device_arguments(hip, oracle):
    """dst on the host, src on the device and vice versa"""
    import torch
    img = Oracle.synth(640, 480, 8)
    dev = torch.from_numpy(img).cuda()
    out = np.full_like(img, SENT)
    hip.blur(out, dev, 3)
    assert_same(out, oracle.blur(img, 3), "host dst, device src")
    dout = torch.full_like(dev, SENT)
    hip.sobel(dout, img)
    assert_same(dout.cpu().numpy(), oracle.sobel(img, np.full_like(img, SENT)), "device dst, host src")
    out = np.full_like(img, SENT)
    hip.sobel(out, dev)
    assert_same(out, oracle.sobel(img, np.full_like(img, SENT)), "host dst (frame kept), device src")


def test_user_stream_and_async_mode(hip, oracle):
    """gsh_set_stream / gsh_set_async: calls enqueue on the caller's stream without a host sync"""
    import torch
    img = Oracle.synth(1280, 720, 4)
    s = torch.cuda.Stream()
    try:
        with torch.cuda.stream(s):
            hip.set_stream(s.cuda_stream)
            hip.set_async(True)
            d = torch.from_numpy(img).cuda()
            a, b = torch.zeros_like(d), torch.zeros_like(d)
            for _ in range(5):
                hip.blur(a, d, 2)
                hip.sobel(b, a)
            s.synchronize()
        assert_same(b.cpu().numpy(), oracle.sobel(oracle.blur(img, 2)), "async chain on a user stream")
    finally:
        hip.set_async(False)
        hip.set_stream(None)


def test_config4_chain_on_one_4k_frame(hip, oracle, cascade):
    """BASELINE configs[4] for ONE frame of its batch: synth(3840x2160, seed 1000) -> gs_blur(2) ->
    gs_sobel (zeroed dst) -> gs_integral -> gs_lbp_detect(frontalface, 4096, 1.1, 1, 4, 1), device
    resident, bit-exact against the oracle (which, like the reference, stops at 4096 detections)."""
    import torch
    w, h = 3840, 2160
    img = Oracle.synth(w, h, 1000)
    src = torch.empty((1, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(src, 1000)
    assert_same(src[0].cpu().numpy(), img, "device-side generator")
    a, b = torch.empty_like(src), torch.zeros_like(src)
    ii = torch.zeros((1, h, w), dtype=torch.int32, device="cuda")
    rects = torch.zeros((1, 4096, 4), dtype=torch.int32, device="cuda")
    counts = torch.zeros(1, dtype=torch.int32, device="cuda")
    ev = torch.zeros(4, dtype=torch.int64, device="cuda")
    dc = hip.cascade_create(cascade)
    hip.blur_batch(a, src, 2)
    hip.sobel_batch(b, a)
    hip.integral_batch(b, ii)
    hip.lbp_count_evaluated(ev)
    try:
        hip.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)
        hip.sync()
    finally:
        hip.lbp_count_evaluated(None)
    dc.close()
    es = oracle.sobel(oracle.blur(img, 2))
    assert_same(b[0].cpu().numpy(), es, "blur -> sobel")
    eii = oracle.integral(es)
    assert_same(ii[0].cpu().numpy().view(np.uint32), eii, "integral of the edge map")
    ro = oracle.lbp_detect(cascade, eii, 4096, 1.1, 1.0, 4.0, 1)
    n = int(counts[0])
    assert n == len(ro) == 4096, (n, len(ro))  # the edge maps of this batch always reach the cap
    got = rects[0, :n].cpu().numpy().view(np.uint32)
    assert_same(got, np.stack([ro["x"], ro["y"], ro["w"], ro["h"]], 1), "first 4096 detections in scan order")
    total = hip.lbp_window_count(cascade, w, h, 1.1, 1.0, 4.0, 1)
    assert total == 120012941
    # on this frame the 4096th detection lies a third of the way into the LAST scale (91x91 windows, y = 658):
    # everything before it has to be evaluated; how many of the chunks behind it are skipped depends on how far the
    # 8 XCD bands of that scale have got when the count is published (with chunks in dispatch order: all of them)
    assert 0.90 * total < int(ev[0]) <= total, "%d of %d windows evaluated" % (int(ev[0]), total)
    print("config4 frame: %d of %d windows evaluated" % (int(ev[0]), total))


def test_lbp_caps_on_edge_maps(hip, oracle, cascade):
    """max_rects = 1, 100, 4096 (and more than there are) on sobel edge maps, host and device tables"""
    import torch
    edges = oracle.sobel(oracle.blur(Oracle.synth(1920, 1080, 1001), 2))
    ii = oracle.integral(edges)
    dii = torch.from_numpy(ii.view(np.int32)).cuda()
    for cap in (1, 100, 4096, 20000):
        ro = oracle.lbp_detect(cascade, ii, cap, 1.1, 1.0, 4.0, 1)
        assert_same(hip.lbp_detect(cascade, ii.copy(), cap, 1.1, 1.0, 4.0, 1), ro, "host table, cap %d" % cap)
        assert_same(hip.lbp_detect(cascade, dii, cap, 1.1, 1.0, 4.0, 1), ro, "device table, cap %d" % cap)
    rc = random_cascade(2)
    ii = oracle.integral(Oracle.synth(1280, 720, 6))
    for cap in (1, 100, 5000, 200000):
        assert_same(hip.lbp_detect(rc, ii.copy(), cap, 1.3, 1.0, 3.0, 1), oracle.lbp_detect(rc, ii, cap, 1.3, 1.0, 3.0, 1),
                    "random cascade, cap %d" % cap)


@pytest.mark.parametrize("knob", [0, 3 + 16 * 10, 15 + 16 * 0, 4 + 16 * 5, 1000 + 4 + 32 * 6 + 1024 * 9])
def test_lbp_adaptive_first_repack(hip, oracle, cascade, knob):
    """the per-block choice of the first survivor re-packing point (default) against forced early / never-early
    choices and a fixed split: same rectangles as the oracle on an edge map and on block noise"""
    import torch
    edges = oracle.sobel(oracle.blur(Oracle.synth(1280, 720, 1002), 2))
    try:
        if knob >= 1000: hip.tune(4, knob)
        elif knob: hip.tune(9, knob)
        for img in (edges, Oracle.synth(1280, 720, 12)):
            ii = oracle.integral(img)
            dii = torch.from_numpy(ii.view(np.int32)).cuda()
            assert_same(hip.lbp_detect(cascade, dii, 4096, 1.1, 1.0, 4.0, 1), oracle.lbp_detect(cascade, ii, 4096, 1.1, 1.0, 4.0, 1),
                        "knob %d" % knob)
    finally:
        hip.tune(4, 0); hip.tune(9, 0)


def test_70000_frame_fast_batch(hip, oracle):
    """more frames than a grid dimension holds (65535) in ONE gsh_fast_batch call"""
    import torch
    from test_emu_logic import fast_many_frames
    n = 70000
    one, ko, smo = fast_many_frames(hip, oracle, n)
    frames = torch.from_numpy(np.repeat(one[None], n, 0)).cuda()
    frames[n - 1] = 5
    sm = torch.zeros_like(frames)
    kps = torch.zeros((n, 4, 12), dtype=torch.int32, device="cuda")
    counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    hip.fast_batch(frames, sm, kps, counts, 4, 20)
    hip.sync()
    c = counts.cpu().numpy()
    assert (c[:n - 1] == 1).all() and c[n - 1] == 0
    k = kps.cpu().numpy().view(np.uint32)
    assert (k[:n - 1, 0] == ko.view(np.uint32).reshape(-1)[None]).all()
    assert bool((sm[:n - 1] == torch.from_numpy(smo).cuda()[None]).all())


def test_two_threads_share_one_cascade_handle(hip, oracle, cascade):
    """gsh_lbp_detect_batch keeps its scan geometry per calling thread, not in the (shared, const)
    handle: two threads scanning different frame sizes through one handle must not disturb each other"""
    import threading
    import torch
    dc = hip.cascade_create(cascade)
    jobs = []
    for (w, h, seed) in ((320, 240, 3), (256, 200, 4)):
        img = Oracle.synth(w, h, seed)
        ii = oracle.integral(img)
        jobs.append((w, h, torch.from_numpy(ii.view(np.int32)).cuda()[None].contiguous(), oracle.lbp_detect(cascade, ii, 500, 1.2, 1.0, 3.0, 1)))
    errors = []

    def work(job):
        try:
            w, h, dii, ro = job
            hip.set_device(0)
            rects = torch.zeros((1, 500, 4), dtype=torch.int32, device="cuda")
            counts = torch.zeros(1, dtype=torch.int32, device="cuda")
            for _ in range(25):
                hip.lbp_detect_batch(dc, dii, rects, counts, 500, 1.2, 1.0, 3.0, 1)
                hip.sync()
                n = int(counts[0])
                got = rects[0, :n].cpu().numpy().view(np.uint32)
                exp = np.stack([ro["x"], ro["y"], ro["w"], ro["h"]], 1) if len(ro) else got
                if n != len(ro) or not np.array_equal(got, exp):
                    errors.append((w, h, n, len(ro)))
                    return
            hip.shutdown()
        except Exception as e:  # noqa
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dc.close()
    assert not errors, errors


def test_histogram_of_a_1_2_gigabyte_image(hip):
    """gs_histogram on one 40000 x 30000 image (1.2e9 bytes): more than one k_hist_partial frame (32-bit offsets), so it is
    counted as one 1 GiB piece + the remainder; against torch.bincount"""
    import torch
    w, h = 40000, 30000
    img = torch.empty((1, h, w), dtype=torch.uint8, device="cuda")
    hip.synth_batch(img[:, :2160, :3840].contiguous(), 3)  # warm the library; the big image is filled by torch below
    torch.manual_seed(1)
    chunk = torch.randint(0, 256, (1000, w), dtype=torch.uint8, device="cuda")
    for y in range(0, h, 1000):
        img[0, y:y + 1000] = chunk.roll(y // 1000, 1)
    img[0, -1, -7:] = 255
    hist = torch.zeros((1, 256), dtype=torch.int32, device="cuda")
    hip.histogram_batch(img, hist)
    ref = torch.zeros(256, dtype=torch.int64, device="cuda")
    for y in range(0, h, 5000):
        ref += torch.bincount(img[0, y:y + 5000].flatten().to(torch.int64), minlength=256)
    assert bool((hist[0].to(torch.int64) == ref).all())
    assert int(hist.sum()) == w * h


@pytest.mark.parametrize("mode", [1, 2, 19])
def test_lbp_chunk_to_xcd_mapping(hip, oracle, cascade, mode):
    """the cascade with chunks in dispatch order (key 13 = 1) and with the XCD-aware mapping forced (2): same rectangles as the
    oracle on a 720p edge map (default: dispatch order at this size) and a 1080p noise frame (default: XCD-aware), caps incl. 5"""
    import torch
    try:
        hip.tune(13, mode)
        for img in (oracle.sobel(oracle.blur(Oracle.synth(1280, 720, 1003), 2)), Oracle.synth(1920, 1080, 13)):
            ii = oracle.integral(img)
            dii = torch.from_numpy(ii.view(np.int32)).cuda()
            for cap in (4096, 5):
                assert_same(hip.lbp_detect(cascade, dii, cap, 1.1, 1.0, 4.0, 1), oracle.lbp_detect(cascade, ii, cap, 1.1, 1.0, 4.0, 1),
                            "mode %d cap %d" % (mode, cap))
    finally:
        hip.tune(13, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [(1, 1), (5, 63), (3, 65), (9, 255), (4, 257), (70, 513), (300, 1030), (2500, 2500)])
def test_match_orb_on_random_descriptors(hip, oracle, n1, n2):
    """k_match reads four train descriptors per lane and trip (round 5): train sets around the trip sizes, ties, near partners"""
    pc.match_random(hip, oracle, n1, n2)
