"""Ragged widths (w % 16 != 0) and frames at any byte address on the strip kernels (round 4).

The API has no stride (ref grayskull.h:14-17), so rows of such frames start at every 16-byte phase; since round 4 the
strip kernels take them (k_strip.h RAGGED / REALIGN) instead of the per-pixel kernels.  Every op is compared bit for
bit with the compiled reference (the pinned restatement where oracle/_ref is absent) inside sentinel-guarded buffers:
a byte written outside the frame, or a wrong byte inside it, fails.

CPU: the kernel sources through the host-fiber emulator (batch entry points take the buffers as "device" memory, so
odd base addresses are real).  GPU (-m gpu): the same cases on the MI355X.
"""
import numpy as np
import pytest

from util import assert_same

GUARD = 192
K3 = np.array([[1, -2, 1], [2, 4, -2], [1, 2, 1]], np.int8)


class Bufs:
    """sentinel-guarded frame batches for one backend"""

    def __init__(self, kind):
        self.kind = kind
        if kind == "gpu":
            import torch
            self.torch = torch

    def make(self, n, h, w, off, fill):
        nb = n * h * w
        if self.kind == "gpu":
            buf = self.torch.full((GUARD + off + nb + GUARD,), 0xAB, dtype=self.torch.uint8, device="cuda")
            v = buf[GUARD + off:GUARD + off + nb].view(n, h, w)
            v.copy_(self.torch.from_numpy(np.ascontiguousarray(fill)))
        else:
            buf = np.full(GUARD + off + nb + GUARD, 0xAB, np.uint8)
            v = buf[GUARD + off:GUARD + off + nb].reshape(n, h, w)
            v[...] = fill
        return buf, v

    def host(self, t):
        return t.cpu().numpy() if self.kind == "gpu" else np.array(t)

    def guards_ok(self, buf, off, nb):
        a = self.host(buf)
        return bool((a[:GUARD + off] == 0xAB).all() and (a[GUARD + off + nb:] == 0xAB).all())


def _ops(g, o):
    return {
        "sobel": (lambda d, s: g.sobel_batch(d, s), lambda img, d0: o.sobel(img, d0)),
        "blur1": (lambda d, s: g.blur_batch(d, s, 1), lambda img, d0: o.blur(img, 1)),
        "blur2": (lambda d, s: g.blur_batch(d, s, 2), lambda img, d0: o.blur(img, 2)),
        "blur3": (lambda d, s: g.blur_batch(d, s, 3), lambda img, d0: o.blur(img, 3)),
        "erode": (lambda d, s: g.erode_batch(d, s), lambda img, d0: o.erode(img)),
        "dilate": (lambda d, s: g.dilate_batch(d, s), lambda img, d0: o.dilate(img)),
        "filter": (lambda d, s: g.filter_batch(d, s, K3, 8), lambda img, d0: o.filter(img, K3, 8)),
    }


def check_width(g, o, bufs, w, h, off, rs, n=2, which=None):
    img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    d0 = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    _, s = bufs.make(n, h, w, off, img)
    for name, (run, ref) in _ops(g, o).items():
        if which and name not in which:
            continue
        db, d = bufs.make(n, h, w, off, d0)
        run(d, s)
        if bufs.kind == "gpu":
            bufs.torch.cuda.synchronize()
        exp = np.stack([ref(img[i], d0[i]) for i in range(n)])
        assert_same(bufs.host(d), exp, "%s %dx%d at base+%d" % (name, w, h, off))
        assert bufs.guards_ok(db, off, n * h * w), "%s %dx%d at base+%d wrote outside the frames" % (name, w, h, off)


def check_more(g, o, bufs, w, h, off, rs, n=2, radii=(4, 16, 17)):
    """the other strip-class ops on the same kind of frames: gs_blur r > 3 / gs_adaptive_threshold (sliding box),
    gs_integral (banded form), gs_downsample"""
    img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    d0 = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    _, s = bufs.make(n, h, w, off, img)
    sync = bufs.torch.cuda.synchronize if bufs.kind == "gpu" else (lambda: None)
    for r in radii:
        db, d = bufs.make(n, h, w, off, d0)
        g.blur_batch(d, s, r), sync()
        assert_same(bufs.host(d), np.stack([o.blur(img[i], r) for i in range(n)]), "blur r=%d %dx%d at base+%d" % (r, w, h, off))
        assert bufs.guards_ok(db, off, n * h * w), "blur r=%d %dx%d wrote outside the frames" % (r, w, h)
        db, d = bufs.make(n, h, w, off, d0)
        g.adaptive_threshold_batch(d, s, r, 3), sync()
        assert_same(bufs.host(d), np.stack([o.adaptive_threshold(img[i], r, 3) for i in range(n)]),
                    "adaptive r=%d %dx%d at base+%d" % (r, w, h, off))
        assert bufs.guards_ok(db, off, n * h * w), "adaptive r=%d %dx%d wrote outside the frames" % (r, w, h)
    # integral: u32 table behind the same kind of guards (4-byte aligned by the C type)
    ib, iv = bufs.make(n, h, 4 * w, 4 * (off // 4) if off >= 4 else 0, np.zeros((n, h, 4 * w), np.uint8))
    ioff = 4 * (off // 4) if off >= 4 else 0
    if bufs.kind == "gpu":
        g.integral_batch(s, iv.view(bufs.torch.int32)), sync()
        got = bufs.host(iv).view(np.uint32)
    else:
        g.integral_batch(s, iv.view(np.uint32))
        got = np.array(iv).view(np.uint32)
    assert_same(got, np.stack([o.integral(img[i]) for i in range(n)]), "integral %dx%d at base+%d" % (w, h, off))
    assert bufs.guards_ok(ib, ioff, n * h * w * 4), "integral %dx%d wrote outside the table" % (w, h)
    if h >= 2:
        hw, hh = w // 2, h // 2
        hb, hv = bufs.make(n, hh, hw, off, np.full((n, hh, hw), 0xCD, np.uint8))
        g.downsample_batch(hv, s), sync()
        assert_same(bufs.host(hv), np.stack([o.downsample(img[i]) for i in range(n)]), "downsample %dx%d at base+%d" % (w, h, off))
        assert bufs.guards_ok(hb, off, n * hh * hw), "downsample %dx%d wrote outside the frames" % (w, h)


def check_fused(g, o, bufs, w, h, off, rs, n=2, radii=(1, 2, 3)):
    """gs_blur -> gs_sobel into a zeroed image in one pass, and the whole config-2 chain behind gsh_edge_pipeline_batch"""
    img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    d0 = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
    _, s = bufs.make(n, h, w, off, img)
    sync = bufs.torch.cuda.synchronize if bufs.kind == "gpu" else (lambda: None)
    for r in radii:
        if h <= 2 * r:
            continue
        exp = np.stack([o.sobel(o.blur(img[i], r), np.zeros((h, w), np.uint8)) for i in range(n)])
        db, d = bufs.make(n, h, w, off, d0)
        g.blur_sobel_batch(d, s, r), sync()
        assert_same(bufs.host(d), exp, "blur+sobel r=%d %dx%d at base+%d" % (r, w, h, off))
        assert bufs.guards_ok(db, off, n * h * w), "blur+sobel r=%d %dx%d wrote outside the frames" % (r, w, h)
        db, d = bufs.make(n, h, w, off, d0)
        if bufs.kind == "gpu":
            hist = bufs.torch.zeros((n, 256), dtype=bufs.torch.int32, device="cuda")
            thr = bufs.torch.zeros(n, dtype=bufs.torch.uint8, device="cuda")
        else:
            hist, thr = np.zeros((n, 256), np.uint32), np.zeros(n, np.uint8)
        g.edge_pipeline_batch(d, None, s, r, hist, thr), sync()
        got, t = bufs.host(d), bufs.host(thr)
        for i in range(n):
            te = o.otsu_threshold(exp[i])
            assert int(t[i]) == te, "otsu %d vs %d (%dx%d r=%d at base+%d)" % (t[i], te, w, h, r, off)
            assert_same(got[i], o.threshold(exp[i], te), "pipeline r=%d %dx%d at base+%d" % (r, w, h, off))
        assert bufs.guards_ok(db, off, n * h * w), "pipeline r=%d %dx%d wrote outside the frames" % (r, w, h)


def _oracle(request):
    from oracle import pyoracle
    return request.getfixturevalue("reference" if pyoracle.have_reference() else "oracle")


def test_emu_all_widths_32_to_1100(emu, request):
    """every width from 32 to 1100: the seam between the grid strips and the anchored tail strip, the idle-lane shift
    (tail in lane 1 of the second wave: w = 1041 .. 1055) and the per-lane edge divisors of gs_blur"""
    o, bufs, rs = _oracle(request), Bufs("emu"), np.random.RandomState(11)
    for w in range(32, 1101):
        check_width(emu, o, bufs, w, 7, 0, rs, n=1)


@pytest.mark.parametrize("off", [1, 2, 3, 4, 8, 15])
def test_emu_frames_at_odd_addresses(emu, request, off):
    o, bufs, rs = _oracle(request), Bufs("emu"), np.random.RandomState(12 + off)
    for w in (32, 48, 61, 64, 100, 612, 1024, 1037, 1041, 1080, 2064, 2071):
        check_width(emu, o, bufs, w, 9, off, rs)


def test_emu_ragged_tall_and_batched(emu, request):
    """bands: several per frame, frames of a batch back to back at every phase"""
    o, bufs, rs = _oracle(request), Bufs("emu"), np.random.RandomState(13)
    emu.tune(0, 5)
    try:
        for w, h in ((37, 41), (612, 33), (1029, 23), (1366, 19)):
            check_width(emu, o, bufs, w, h, 3, rs, n=3)
    finally:
        emu.tune(0, 0)


def test_emu_box_integral_downsample_any_width(emu, request):
    o, bufs, rs = _oracle(request), Bufs("emu"), np.random.RandomState(15)
    for w in list(range(32, 66)) + [100, 255, 257, 612, 1021, 1023, 1025, 1039, 2047, 2049, 4094]:
        check_more(emu, o, bufs, w, 11, 0, rs, n=1)
    for w, off in ((37, 1), (64, 3), (612, 5), (1029, 8), (1366, 15)):
        check_more(emu, o, bufs, w, 23, off, rs, n=2)


def check_ring_box(g, o, bufs, shapes, radii, rs, n=2):
    """gs_blur / gs_adaptive_threshold with 4 <= r <= 16 on ragged rows tall enough for a whole window (the radii the
    register-ring kernels take: on ragged rows the whole strips with the last one a feeder + k_box_edge for the last
    16 + w % 16 columns): a bright right edge, where a wrong edge divisor would show, positive and negative c"""
    sync = bufs.torch.cuda.synchronize if bufs.kind == "gpu" else (lambda: None)
    for (w, h, off) in shapes:
        img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
        img[:, :, w - 20:] = rs.randint(200, 256, (n, h, 20))
        d0 = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
        _, s = bufs.make(n, h, w, off, img)
        for r in radii:
            if h < 2 * r + 1:
                continue
            db, d = bufs.make(n, h, w, off, d0)
            g.blur_batch(d, s, r), sync()
            assert_same(bufs.host(d), np.stack([o.blur(img[i], r) for i in range(n)]), "blur r=%d %dx%d at base+%d" % (r, w, h, off))
            assert bufs.guards_ok(db, off, n * h * w), "blur r=%d %dx%d wrote outside the frames" % (r, w, h)
            for c in (3, -7):
                db, d = bufs.make(n, h, w, off, d0)
                g.adaptive_threshold_batch(d, s, r, c), sync()
                assert_same(bufs.host(d), np.stack([o.adaptive_threshold(img[i], r, c) for i in range(n)]),
                            "adaptive r=%d c=%d %dx%d at base+%d" % (r, c, w, h, off))
                assert bufs.guards_ok(db, off, n * h * w), "adaptive r=%d %dx%d wrote outside the frames" % (r, w, h)


def test_emu_ring_box_on_ragged_rows(emu, request):
    o, bufs, rs = _oracle(request), Bufs("emu"), np.random.RandomState(17)
    check_ring_box(emu, o, bufs, ((33, 40, 0), (47, 35, 1), (49, 70, 0), (100, 33, 3), (1039, 34, 0), (2049, 36, 2)), (4, 5, 8, 11, 16), rs)


def test_emu_integral_wider_than_4096(emu, request):
    """column chunks of 4096 px: a row's prefix at the chunk edge waits in a lane register (k_integral_wave WIDE)"""
    o, rs = _oracle(request), np.random.RandomState(16)
    for w, h in ((4097, 9), (5000, 70), (8192, 33), (8193, 5), (9001, 130)):
        img = rs.randint(0, 256, (2, h, w)).astype(np.uint8)
        ii = np.full((2, h, w), 0xABABABAB, np.uint32)
        emu.integral_batch(img, ii)
        assert_same(ii, np.stack([o.integral(img[i]) for i in range(2)]), "integral %dx%d" % (w, h))


def test_emu_fused_blur_sobel_any_width(emu, request):
    o, bufs, rs = _oracle(request), Bufs("emu"), np.random.RandomState(17)
    for w in list(range(32, 70)) + [100, 255, 257, 612, 1009, 1023, 1025, 1039, 1041, 1055, 2047, 2065]:
        check_fused(emu, o, bufs, w, 9, 0, rs, n=1)
    for w, off in ((37, 1), (64, 3), (612, 5), (1029, 8), (1366, 15)):
        check_fused(emu, o, bufs, w, 40, off, rs, n=2)


def test_emu_fast_any_width(emu, request):
    """gs_fast pass 2 on the strip machinery for any width (the strips stay on the grid, flags are masked)"""
    o, rs = _oracle(request), np.random.RandomState(14)
    for w in list(range(32, 70)) + [100, 127, 129, 612, 1000, 1023, 1025, 1041, 1080]:
        for h in (9, 20):
            img = (rs.randint(0, 256, (h, w)) * (rs.rand(h, w) < 0.7)).astype(np.uint8)
            sm0 = rs.randint(0, 256, (h, w)).astype(np.uint8)
            sm = sm0.copy()
            k = emu.fast(img, sm, 5000, 20)
            ko, smo = o.fast(img, 5000, 20, sm0)
            assert_same(k, ko, "fast %dx%d" % (w, h))
            assert_same(sm, smo, "fast scoremap %dx%d" % (w, h))


@pytest.mark.gpu
def test_gpu_all_widths_32_to_1100(hip, request):
    o, bufs, rs = _oracle(request), Bufs("gpu"), np.random.RandomState(21)
    for w in range(32, 1101):
        check_width(hip, o, bufs, w, 7, 0, rs, n=1)


@pytest.mark.gpu
@pytest.mark.parametrize("off", [1, 2, 3, 4, 8, 15])
def test_gpu_frames_at_odd_addresses(hip, request, off):
    o, bufs, rs = _oracle(request), Bufs("gpu"), np.random.RandomState(22 + off)
    for w in (32, 48, 61, 64, 100, 612, 1024, 1037, 1041, 1080, 2064, 2071, 3838, 3840):
        check_width(hip, o, bufs, w, 23, off, rs, n=3)


@pytest.mark.gpu
def test_gpu_ring_box_on_ragged_rows(hip, request):
    o, bufs, rs = _oracle(request), Bufs("gpu"), np.random.RandomState(27)
    check_ring_box(hip, o, bufs, ((33, 40, 0), (47, 35, 1), (100, 33, 3), (612, 120, 5), (1039, 64, 0), (1366, 70, 15), (3838, 40, 2)),
                   (4, 9, 16), rs)
    try:  # k_box_edge on the side stream whatever the batch size and radius (the rule: r >= 12 on batches of 16 Mpx and more)
        hip.tune(6, 7)
        check_ring_box(hip, o, bufs, ((47, 35, 1), (612, 120, 5), (1366, 70, 15), (3838, 70, 2)), (5, 16), rs)
    finally:
        hip.tune(6, 0)
    # the rule's own choice on a batch large enough for the side stream: 9 x 1918 x 1080 = 18.6 Mpx
    check_ring_box(hip, o, bufs, ((1918, 1080, 0),), (13,), rs, n=9)


@pytest.mark.gpu
def test_gpu_box_integral_downsample_any_width(hip, request):
    o, bufs, rs = _oracle(request), Bufs("gpu"), np.random.RandomState(24)
    for w in list(range(32, 66)) + [100, 255, 257, 612, 1021, 1023, 1025, 1039, 2047, 2049, 4094]:
        check_more(hip, o, bufs, w, 11, 0, rs, n=1)
    for w, h, off in ((37, 23, 1), (64, 23, 3), (612, 816, 5), (1029, 40, 8), (1366, 768, 15), (1080, 1920, 0), (3838, 300, 2)):
        check_more(hip, o, bufs, w, h, off, rs, n=2)


@pytest.mark.gpu
def test_gpu_fused_blur_sobel_any_width(hip, request):
    o, bufs, rs = _oracle(request), Bufs("gpu"), np.random.RandomState(26)
    for w in list(range(32, 70)) + [100, 255, 257, 612, 1009, 1023, 1025, 1039, 1041, 1055, 2047, 2065]:
        check_fused(hip, o, bufs, w, 9, 0, rs, n=1)
    for w, h, off in ((37, 40, 1), (64, 40, 3), (612, 816, 5), (1080, 1920, 0), (1366, 768, 15), (3838, 2160, 0), (3840, 2160, 1)):
        check_fused(hip, o, bufs, w, h, off, rs, n=2, radii=(2,) if w * h > 10**6 else (1, 2, 3))


@pytest.mark.gpu
def test_gpu_integral_wider_than_4096(hip, request):
    import torch
    o, rs = _oracle(request), np.random.RandomState(25)
    for w, h in ((4097, 9), (5000, 70), (8192, 33), (7680, 4320), (9001, 130)):
        n = 1 if w * h > 10**7 else 2
        img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
        ii = torch.full((n, h, w), -1, dtype=torch.int32, device="cuda")
        hip.integral_batch(torch.from_numpy(img).cuda(), ii)
        assert_same(ii.cpu().numpy().view(np.uint32), np.stack([o.integral(img[i]) for i in range(n)]), "integral %dx%d" % (w, h))


@pytest.mark.gpu
def test_gpu_ragged_video_sizes(hip, request):
    """the reference's own fixture size (testdata/receipt.pgm, 612 x 816), portrait 1080p, 1366 x 768, a cropped 4K"""
    o, bufs, rs = _oracle(request), Bufs("gpu"), np.random.RandomState(23)
    for w, h in ((612, 816), (1080, 1920), (1366, 768), (3838, 2160)):
        check_width(hip, o, bufs, w, h, 0, rs, n=2)
        check_width(hip, o, bufs, w, h, 1, rs, n=1, which=("sobel", "blur2"))
