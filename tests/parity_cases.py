"""Backend-agnostic parity checks: HIP path (or its emulated kernel logic) vs the CPU oracle.

`g`   : grayskull_amd.Grayskull bound to a library
`o`   : oracle.pyoracle.Oracle
`mem` : Mem("host") -> numpy arrays (staged path) or Mem("device") -> torch CUDA tensors
        (zero-copy path)
Integer/byte/index outputs are compared bit-exact; the only float output (keypoint angle) is
also compared bit-exact because the angle comes from the same host libm as the oracle's.
"""
import numpy as np

from grayskull_amd import KEYPOINT_DTYPE, MATCH_DTYPE, RECT_DTYPE
from oracle.pyoracle import Oracle
from util import assert_same

SENTINEL = 0xAB


class Mem:
    def __init__(self, kind="host"):
        self.kind = kind

    def put(self, a):
        a = np.ascontiguousarray(a)
        if self.kind == "host":
            return a.copy()
        import torch
        if a.dtype == np.uint32:  # torch has no uint32 arithmetic; carry the bits as int32
            return torch.from_numpy(a.view(np.int32).copy()).cuda()
        return torch.from_numpy(a.copy()).cuda()

    def get(self, t, dtype=None):
        if self.kind == "host":
            return t
        a = t.cpu().numpy()
        return a.view(dtype) if dtype is not None else a

    def zeros(self, shape, dtype=np.uint8, fill=0):
        return self.put(np.full(shape, fill, dtype))


def orb_batch(g, o, frames, mem, nkps=40, threshold=20):
    """gsh_orb_extract_batch (n same-size frames, two round trips) == gs_orb_extract per frame"""
    sm0 = np.random.RandomState(7).randint(0, 256, frames.shape).astype(np.uint8)
    got = g.orb_extract_batch_dev(mem.put(frames), mem.put(sm0), nkps, threshold)
    assert len(got) == len(frames)
    for f, img in enumerate(frames):
        assert_same(got[f], o.orb_extract(img, nkps, threshold, sm0[f]), "orb batch frame %d" % f)


def orb_pyramid(g, o, img, mem, nkps=90, threshold=20, levels=3, seed=1):
    """gsh_orb_extract_pyramid (nanomagick.c:245-290 with device-resident levels) vs the oracle,
    with a non-zero scratch buffer: pyramid levels, scoremaps and keypoints must all match"""
    h, w = img.shape
    nb = g.orb_pyramid_buffer_bytes(w, h, levels)
    assert nb == o.orb_pyramid_buffer_bytes(w, h, levels)
    buf0 = np.random.RandomState(seed).randint(0, 256, nb + 16).astype(np.uint8)
    ko, bo = o.orb_extract_pyramid(img, nkps, threshold, levels, buf0)
    buf = mem.put(buf0)
    k = g.orb_extract_pyramid_dev(mem.put(img), buf, nkps, threshold, levels)
    assert len(k) == len(ko), "pyramid ORB count %d vs %d" % (len(k), len(ko))
    assert_same(k, ko, "pyramid ORB keypoints")
    assert_same(mem.get(buf)[:nb], bo[:nb], "pyramid levels + scoremaps")


def geometry(g, o, img, mem, seed=3):
    """gs_crop/gs_copy, gs_resize_nn, gs_resize (float32 bilinear, bit-exact),
    gs_match_template + gs_find_best_match"""
    h, w = img.shape
    s = mem.put(img)
    rs = np.random.RandomState(seed)
    for (rx, ry, rw, rh) in ((0, 0, w, h), (1, 0, max(w - 1, 1), h), (w // 3, h // 2, max(w // 2, 1), max(h // 3, 1))):
        if rx + rw > w or ry + rh > h:
            continue
        d = mem.zeros((rh, rw), fill=SENTINEL)
        g.crop(d, s, rx, ry, rw, rh)
        assert_same(mem.get(d), o.crop(img, rx, ry, rw, rh), "gs_crop %s" % ((rx, ry, rw, rh),))
    d = mem.zeros((h, w), fill=SENTINEL)
    g.copy(d, s)
    assert_same(mem.get(d), img, "gs_copy")
    for (dw, dh) in ((2 * w, 2 * h), (w // 2 + 1, h // 2 + 1), (w, h), (13, 7), (3 * w + 1, 2)):
        for nn in (False, True):
            d = mem.zeros((dh, dw), fill=SENTINEL)
            g.resize(d, s, nn)
            assert_same(mem.get(d), o.resize(img, dw, dh, nn), "gs_resize%s -> %dx%d" % ("_nn" if nn else "", dw, dh))
    tmpls = [rs.randint(0, 256, (th, tw)).astype(np.uint8) for (tw, th) in ((3, 3), (max(w // 2, 1), max(h // 2, 1)), (1, 1), (w, h))]
    if w > 12 and h > 10:
        tmpls.append(img[2:10, 3:11].copy())  # an exact sub-image: best match at (3, 2) unless repeated
    for t in tmpls:
        th, tw = t.shape
        r = mem.zeros((h - th + 1, w - tw + 1), fill=SENTINEL)
        g.match_template(s, mem.put(t), r)
        ro = o.match_template(img, t)
        assert_same(mem.get(r), ro, "gs_match_template %dx%d" % (tw, th))
        assert g.find_best_match(r) == o.find_best_match(ro), "gs_find_best_match"
    z = mem.zeros((5, 7), fill=0)
    assert g.find_best_match(z) == (0, 0), "gs_find_best_match on all zeros"


def stencils(g, o, img, mem, radii=(1, 2, 3, 5)):
    s = mem.put(img)
    for r in radii:
        d = mem.zeros(img.shape, fill=SENTINEL)
        g.blur(d, s, r)
        assert_same(mem.get(d), o.blur(img, r), "gs_blur r=%d %s" % (r, img.shape))
    d = mem.zeros(img.shape, fill=SENTINEL)
    g.sobel(d, s)
    assert_same(mem.get(d), o.sobel(img, np.full_like(img, SENTINEL)),
                "gs_sobel (1-px frame must keep the sentinel) %s" % (img.shape,))
    # ... and whatever the caller had there: a dst whose border differs from row to row (a constant
    # sentinel cannot tell rows apart -- a kernel that restored column 0 from the wrong row passed it)
    d0 = np.random.RandomState(img.shape[0] * 7 + img.shape[1]).randint(0, 256, img.shape).astype(np.uint8)
    d = mem.put(d0)
    g.sobel(d, s)
    assert_same(mem.get(d), o.sobel(img, d0), "gs_sobel (random dst: frame untouched) %s" % (img.shape,))
    for name in ("erode", "dilate"):
        d = mem.zeros(img.shape, fill=SENTINEL)
        getattr(g, name)(d, s)
        assert_same(mem.get(d), getattr(o, name)(img), "gs_%s %s" % (name, img.shape))


def pointwise(g, o, img, mem):
    s = mem.put(img)
    assert_same(g.histogram(s), o.histogram(img), "gs_histogram")
    assert g.otsu_threshold(s) == o.otsu_threshold(img), "gs_otsu_threshold"
    for t in (0, 1, 100, 254, 255, o.otsu_threshold(img)):
        d = mem.put(img)
        g.threshold(d, t)
        assert_same(mem.get(d), o.threshold(img, t), "gs_threshold t=%d" % t)


def integral(g, o, img, mem):
    s = mem.put(img)
    if mem.kind == "host":
        ii = g.integral(s)
    else:
        ii_t = mem.put(np.zeros(img.shape, np.uint32))
        g.integral(s, ii_t)
        ii = mem.get(ii_t, np.uint32)
    assert_same(ii, o.integral(img), "gs_integral")


def next_rows(g, o, img, mem):
    s = mem.put(img)
    big = img.size > 60000  # the CPU side of a radius-100 box is ~40,000 taps per pixel: big radii on small images only
    # c incl. values where px + c leaves [0, 256] and where the reference's unsigned `mean - c` wraps (|c| >= 2^30: literal path)
    for (r, c) in ((1, 0), (3, 5), (15, 5), (2, -7), (4, 300), (5, -300), (4, 255), (6, -255), (4, 256), (9, 1 << 30), (4, -(1 << 31)),
                   (7, (1 << 31) - 1), (5, -(1 << 30) + 1), (8, (1 << 30) - 1), (57, 5), (100, -4), (127, 9), (128, 0)):
        if big and r > 16:
            continue
        d = mem.zeros(img.shape, fill=SENTINEL)
        g.adaptive_threshold(d, s, r, c)
        assert_same(mem.get(d), o.adaptive_threshold(img, r, c), "gs_adaptive_threshold r=%d c=%d" % (r, c))
    kernels = {"sharpen": ([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], 1),
               "emboss": ([[-2, -1, 0], [-1, 1, 1], [0, 1, 2]], 1),
               "box": ([[1, 1, 1], [1, 1, 1], [1, 1, 1]], 9),
               "gauss": ([[1, 2, 1], [2, 4, 2], [1, 2, 1]], 16),
               "neg_norm": ([[0, -1, 0], [-1, 2, -1], [0, -1, 0]], 3),
               "wide": ([[1, 0, -1, 2, 1]], 2),
               # strip-kernel boundaries: sum |k| == 128 (largest int16-safe), 129 (falls back), norms 255/256/257
               "abs128": ([[16, -16, 16], [-16, 0, 16], [16, -16, 16]], 7),
               "abs129": ([[16, -16, 16], [-16, 1, 16], [16, -16, 16]], 7),
               "all_pos_128": ([[14, 14, 14], [14, 16, 14], [14, 14, 14]], 1),
               "norm255": ([[14, 14, 14], [14, 16, 14], [14, 14, 14]], 255),
               "norm256": ([[14, 14, 14], [14, 16, 14], [14, 14, 14]], 256),
               "norm257": ([[14, 14, 14], [14, 16, 14], [14, 14, 14]], 257),
               "all_neg": ([[-1, -2, -1], [-2, -4, -2], [-1, -2, -1]], 1),
               "all_neg_norm": ([[-1, -2, -1], [-2, -4, -2], [-1, -2, -1]], 16),
               "min_int8": ([[0, 0, 0], [0, -128, 0], [0, 0, 0]], 2)}
    for name, (k, norm) in kernels.items():
        k = np.array(k, np.int8)
        d = mem.zeros(img.shape, fill=SENTINEL)
        g.filter(d, s, k, norm)
        assert_same(mem.get(d), o.filter(img, k, norm), "gs_filter %s" % name)
    if img.shape[0] >= 2 and img.shape[1] >= 2:
        d = mem.zeros((img.shape[0] // 2, img.shape[1] // 2), fill=SENTINEL)
        g.downsample(d, s)
        assert_same(mem.get(d), o.downsample(img), "gs_downsample")


def fast(g, o, img, mem, threshold=20, caps=(5000, 7)):
    h, w = img.shape
    rs = np.random.RandomState(w * 131 + h)
    sm0 = rs.randint(0, 256, (h, w)).astype(np.uint8)  # caller-owned frame content is read by NMS
    s = mem.put(img)
    for cap in caps:
        sm = mem.put(sm0)
        k = g.fast(s, sm, cap, threshold)
        ko, smo = o.fast(img, cap, threshold, sm0)
        assert_same(k, ko, "gs_fast keypoints cap=%d" % cap)
        assert_same(mem.get(sm), smo, "gs_fast scoremap (3-px frame untouched)")


def fast_unsigned_wrap_quirk(g, o, mem):
    """SURVEY 8c quirk KAT: p < threshold makes the 'darker' bound wrap (grayskull.h:498)"""
    img = np.full((9, 9), 5, np.uint8)
    for dx, dy in zip((0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1),
                      (-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3)):
        img[4 + dy, 4 + dx] = 20
    sm = mem.put(np.zeros_like(img))
    k = g.fast(mem.put(img), sm, 16, 20)
    ko, smo = o.fast(img, 16, 20)
    assert_same(k, ko, "FAST wrap quirk keypoints")
    assert len(k) == 1 and k[0]["response"] == 15 and mem.get(sm)[4, 4] == 15


def orb(g, o, img, mem, nkps=50, threshold=20):
    h, w = img.shape
    s = mem.put(img)
    sm = mem.put(np.zeros_like(img))
    k = g.orb_extract(s, nkps, threshold, sm)
    ko = o.orb_extract(img, nkps, threshold)
    assert len(k) == len(ko), "gs_orb_extract count %d vs %d" % (len(k), len(ko))
    for f in ("x", "y", "response", "desc"):
        assert_same(k[f], ko[f], "gs_orb_extract." + f)
    # angle tolerance per north_star is 1e-5; same host libm => we also expect identical bits
    assert np.allclose(k["angle"], ko["angle"], rtol=0, atol=1e-5), "gs_orb_extract.angle"
    assert_same(k["angle"].view(np.uint32), ko["angle"].view(np.uint32), "gs_orb_extract.angle bits")
    # shifted second view -> matching
    B = np.zeros_like(img)
    B[:h - 3, :w - 5] = img[3:, 5:]
    kb = g.orb_extract(mem.put(B), nkps, threshold, mem.put(np.zeros_like(img)))
    kbo = o.orb_extract(B, nkps, threshold)
    assert_same(kb, kbo, "gs_orb_extract (shifted frame)")
    for (mm, md) in ((4 * nkps, 60.0), (3, 80.0), (4 * nkps, 0.0), (4 * nkps, 256.0)):
        m = g.match_orb(k, kb, mm, md)
        mo = o.match_orb(ko, kbo, mm, md)
        assert_same(m, mo, "gs_match_orb max=%d dist=%g" % (mm, md))
    # single-keypoint API
    for (x, y) in ((20, 20), (w - 16, h - 16), (15, 15)):
        if x >= 15 and y >= 15 and x < w - 15 and y < h - 15:
            a, ao = g.compute_orientation(s, x, y, 15), o.orientation(img, x, y, 15)
            assert np.float32(a).view(np.uint32) == np.float32(ao).view(np.uint32), "gs_compute_orientation"
    for (x, y, ang) in ((30, 22, 0.7), (3, 2, -2.1), (w - 1, h - 1, 3.0), (w // 2, h // 2, 0.0)):
        assert_same(g.brief_descriptor(s, x, y, ang), o.brief(img, x, y, ang), "gs_brief_descriptor")


def lbp(g, o, img, mem, casc, params=((4096, 1.1, 1.0, 4.0, 1),), windows=((0, 0, 1.0),)):
    ii = o.integral(img)
    ii_m = mem.put(ii)
    for (mr, sf, mn, mx, step) in params:
        r = g.lbp_detect(casc, ii_m, mr, sf, mn, mx, step)
        ro = o.lbp_detect(casc, ii, mr, sf, mn, mx, step)
        assert_same(r, ro, "gs_lbp_detect max=%d sf=%g [%g,%g] step=%d" % (mr, sf, mn, mx, step))
    for (x, y, sc) in windows:
        assert g.lbp_window(casc, ii_m, x, y, sc) == o.lbp_window(casc, ii, x, y, sc), \
            "gs_lbp_window (%d,%d,%g)" % (x, y, sc)


def orb_nostdlib(g, o_nostdlib, frames, nkps=60, threshold=20):
    """gsh_orb_extract_batch_nostdlib (device-resident, GS_NO_STDLIB trig of ref :70-88) vs the reference header
    compiled with -DGS_NO_STDLIB (or the restatement switched to the same polynomials), frame by frame"""
    frames = np.ascontiguousarray(frames)
    n = frames.shape[0]
    mem = Mem("device") if _is_gpu(g) else Mem("host")
    d = mem.put(frames)
    sm = mem.zeros(frames.shape)
    kps = mem.put(np.zeros((n, nkps, 12), np.uint32))
    counts = mem.put(np.zeros(n, np.uint32))
    g.orb_extract_batch_nostdlib(d, sm, kps, counts, nkps, threshold)
    g.sync()
    k, c = mem.get(kps, np.uint32), mem.get(counts, np.uint32)
    for f in range(n):
        ko = o_nostdlib.orb_extract(frames[f], nkps, threshold)
        assert int(c[f]) == len(ko), "frame %d: %d vs %d keypoints" % (f, int(c[f]), len(ko))
        assert_same(k[f, :len(ko)].reshape(-1).view(ko.dtype), ko, "device-resident ORB (NO_STDLIB trig) frame %d" % f)


def _is_gpu(g):
    return "emulator" not in g.version()


def match_random(g, o, n1, n2):
    """gs_match_orb on random descriptors: partners at distance 0 / 3 / 40, some of them twice in the train set (ties: the first
    index of the minimum wins, ref :690; equal best and second fail the 0.8 ratio test), three (max_matches, max_distance)"""
    from grayskull_amd import KEYPOINT_DTYPE
    rng = np.random.RandomState(1000 * n1 + n2)
    k1, k2 = np.zeros(n1, KEYPOINT_DTYPE), np.zeros(n2, KEYPOINT_DTYPE)
    raw1, raw2 = k1.view(np.uint32).reshape(n1, 12), k2.view(np.uint32).reshape(n2, 12)
    raw1[:, 4:] = rng.randint(0, 2 ** 32, (n1, 8), dtype=np.uint64).astype(np.uint32)
    raw2[:, 4:] = rng.randint(0, 2 ** 32, (n2, 8), dtype=np.uint64).astype(np.uint32)
    for i in range(n1):
        j = int(rng.randint(0, n2))
        raw2[j, 4:] = raw1[i, 4:]
        if i % 3 == 1:
            raw2[j, 4] ^= 0x7
        if i % 3 == 2:
            raw2[j, 5] ^= 0xFFFFF00F
            raw2[j, 6] ^= 0xFFFF
        if i % 2 and n2 > 2:
            raw2[(j + n2 // 2) % n2, 4:] = raw2[j, 4:]
    for mm, md in ((n1 + 3, 60.0), (max(1, n1 // 2), 256.0), (n1, 2.0)):
        assert_same(g.match_orb(k1, k2, mm, md), o.match_orb(k1, k2, mm, md), "gs_match_orb %d x %d max=%d dist=%g" % (n1, n2, mm, md))
