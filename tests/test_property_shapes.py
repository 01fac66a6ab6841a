"""Property test (hypothesis): for random small shapes -- biased towards the strip-kernel boundaries
(w a multiple of 16, tiny heights, radius >= height) -- every drop-in function run through the
kernel sources (host-fiber emulator) equals the oracle.  Complements the fixed shape lists."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import parity_cases as pc
from util import assert_same

MEM = pc.Mem("host")
# Deterministic in CI (the same examples every run); GS_HYPOTHESIS_EXAMPLES=N explores N fresh random
# examples per test instead (used while developing: it found the gs_sobel dst-border bug).
import os
_N = int(os.environ.get("GS_HYPOTHESIS_EXAMPLES", "0"))


def _cfg(default):
    return settings(max_examples=_N or default, deadline=None, derandomize=not _N, database=None,
                    suppress_health_check=list(HealthCheck))
widths = st.one_of(st.sampled_from([16, 32, 48, 64, 80, 96, 112, 1024, 1040]), st.integers(1, 70))
heights = st.one_of(st.integers(1, 12), st.integers(13, 70))


def _img(rs, w, h, kind):
    if kind == 0:
        return rs.randint(0, 256, (h, w)).astype(np.uint8)
    if kind == 1:  # flat blocks + small noise: many equal neighbours, clamps rarely hit
        return np.clip(rs.randint(0, 256) + rs.randint(-3, 4, (h, w)), 0, 255).astype(np.uint8)
    return (rs.randint(0, 2, (h, w)) * 255).astype(np.uint8)  # binary: saturating sums


def _body_stencils_any_shape(emu, oracle, w, h, seed, kind, radius):
    if w >= 1024 and h > 8:
        h = 8  # keep the emulator fast
    img = _img(np.random.RandomState(seed), w, h, kind)
    s = img.copy()
    d = np.full_like(img, 0xAB)
    emu.blur(d, s, radius)
    assert_same(d, oracle.blur(img, radius), "gs_blur r=%d %dx%d" % (radius, w, h))
    for name in ("erode", "dilate"):
        d = np.full_like(img, 0xAB)
        getattr(emu, name)(d, s)
        assert_same(d, getattr(oracle, name)(img), "gs_%s %dx%d" % (name, w, h))
    if w >= 3 and h >= 3:
        d0 = np.random.RandomState(seed + 1).randint(0, 256, (h, w)).astype(np.uint8)
        d = d0.copy()
        emu.sobel(d, s)
        assert_same(d, oracle.sobel(img, d0), "gs_sobel %dx%d (frame of dst kept)" % (w, h))
    d = np.full_like(img, 0xAB)
    emu.adaptive_threshold(d, s, radius, 5)
    assert_same(d, oracle.adaptive_threshold(img, radius, 5), "gs_adaptive_threshold r=%d %dx%d" % (radius, w, h))
    assert_same(emu.integral(s), oracle.integral(img), "gs_integral %dx%d" % (w, h))
    assert_same(emu.histogram(s), oracle.histogram(img), "gs_histogram")
    assert emu.otsu_threshold(s) == oracle.otsu_threshold(img)
    t = (seed * 7) % 256
    c = img.copy()
    emu.threshold(c, t)
    assert_same(c, oracle.threshold(img, t), "gs_threshold t=%d" % t)
    cneg = -((seed % 50) + 1)
    d = np.full_like(img, 0xAB)
    emu.adaptive_threshold(d, s, radius, cneg)
    assert_same(d, oracle.adaptive_threshold(img, radius, cneg), "gs_adaptive_threshold c=%d" % cneg)


def _body_filter_any_shape(emu, oracle, w, h, seed, norm, ks):
    if w >= 1024 and h > 8:
        h = 8
    img = _img(np.random.RandomState(seed), w, h, seed % 3)
    k = np.array(ks, np.int8).reshape(3, 3)
    d = np.full_like(img, 0xAB)
    emu.filter(d, img.copy(), k, norm)
    assert_same(d, oracle.filter(img, k, norm), "gs_filter %dx%d norm=%d k=%s" % (w, h, norm, ks))


def _body_fused_pipeline_any_shape(emu, oracle, n, w, h, radius, seed):
    if w >= 1024 and h > 8:
        h = 8
    rs = np.random.RandomState(seed)
    src = np.stack([_img(rs, w, h, i % 3) for i in range(n)])
    mem = pc.Mem("host" if "kernel_emu" in emu.path else "device")  # the batch API takes device memory
    out_m, hist_m, thr_m = mem.put(np.full_like(src, 7)), mem.put(np.zeros((n, 256), np.uint32)), mem.put(np.zeros(n, np.uint8))
    emu.edge_pipeline_batch(out_m, None, mem.put(src), radius, hist_m, thr_m)
    if mem.kind == "device":
        emu.sync()
    out, thr = mem.get(out_m), mem.get(thr_m)
    for i in range(n):
        s = oracle.sobel(oracle.blur(src[i], radius))
        t = oracle.otsu_threshold(s)
        assert int(thr[i]) == t, "otsu frame %d" % i
        assert_same(out[i], oracle.threshold(s, t), "pipeline frame %d (%dx%d r=%d)" % (i, w, h, radius))


def _body_fast_orb_match_any_shape(emu, oracle, w, h, seed, kind, threshold, nkps):
    rs = np.random.RandomState(seed)
    img = _img(rs, w, h, kind)
    sm0 = rs.randint(0, 256, (h, w)).astype(np.uint8)  # the caller's scoremap frame is read by the NMS
    sm = sm0.copy()
    k = emu.fast(img.copy(), sm, nkps, threshold)
    ko, smo = oracle.fast(img, nkps, threshold, sm0)
    assert_same(k, ko, "gs_fast %dx%d t=%d cap=%d" % (w, h, threshold, nkps))
    assert_same(sm, smo, "gs_fast scoremap")
    ka = emu.orb_extract(img.copy(), nkps, threshold, sm0.copy())
    kao = oracle.orb_extract(img, nkps, threshold, sm0)
    assert_same(ka, kao, "gs_orb_extract %dx%d t=%d n=%d" % (w, h, threshold, nkps))
    if len(ka):
        B = np.roll(img, (1, 2), (0, 1))
        kb, kbo = emu.orb_extract(B.copy(), nkps, threshold, sm0.copy()), oracle.orb_extract(B, nkps, threshold, sm0)
        assert_same(kb, kbo, "gs_orb_extract (second frame)")
        for (mm, md) in ((2 * nkps, 64.0), (3, 256.0)):
            assert_same(emu.match_orb(ka, kb, mm, md), oracle.match_orb(kao, kbo, mm, md), "gs_match_orb")


def _body_fast_px_kernel(emu, oracle, w4, h, seed, kind, threshold, dark):
    """gsh_tune key 7 = 2: k_fast_score_px + the item-by-item NMS (the generic pair behind huge thresholds); widths around one and
    two 256-px spans"""
    rs = np.random.RandomState(seed)
    w = 4 * w4
    img = _img(rs, w, h, kind)
    if dark:  # a p < threshold region: the reference's unsigned wrap class
        img[h // 3: 2 * h // 3, w // 4: w // 2] = rs.randint(0, max(2, min(threshold, 255)), (2 * h // 3 - h // 3, w // 2 - w // 4))
    sm0 = rs.randint(0, 256, (h, w)).astype(np.uint8)
    ko, smo = oracle.fast(img, 5000, threshold, sm0)
    emu.tune(7, 2)
    try:
        sm = sm0.copy()
        k = emu.fast(img.copy(), sm, 5000, threshold)
    finally:
        emu.tune(7, 0)
    assert_same(k, ko, "gs_fast (per-pixel kernel) %dx%d t=%d" % (w, h, threshold))
    assert_same(sm, smo, "gs_fast scoremap (per-pixel kernel)")


def _body_fast_wide(emu, oracle, w, h, seed, kind, threshold, cap, key19):
    """gs_fast on frames several score tiles wide and tall (64 x 48-px tiles, bitmap words, chunks of 32 words that straddle
    rows), a caller's non-zero score map, caps that cut the list; key19: 0 sparse NMS, 1 item by item"""
    rs = np.random.RandomState(seed)
    img = _img(rs, w, h, kind)
    if kind == 1:  # p < threshold regions: every pixel a candidate under the reference's unsigned wrap
        img[h // 4: h // 2, w // 3: 2 * w // 3] = rs.randint(0, max(2, min(threshold, 255)), (h // 2 - h // 4, 2 * w // 3 - w // 3))
    sm0 = rs.randint(0, 256, (h, w)).astype(np.uint8)
    ko, smo = oracle.fast(img, cap, threshold, sm0)
    emu.tune(19, key19)
    try:
        sm = sm0.copy()
        k = emu.fast(img.copy(), sm, cap, threshold)
    finally:
        emu.tune(19, 0)
    assert_same(k, ko, "gs_fast %dx%d t=%d cap=%d key19=%d" % (w, h, threshold, cap, key19))
    assert_same(sm, smo, "gs_fast scoremap %dx%d" % (w, h))


def _body_lbp_any_shape(emu, oracle, w, h, seed, cseed, sf, mx, step, cap):
    from util import random_cascade
    img = _img(np.random.RandomState(seed), w, h, seed % 3)
    casc = random_cascade(cseed, nstages=2 + cseed % 3, weaks_per_stage=1 + cseed % 4)
    ii = oracle.integral(img)
    r = emu.lbp_detect(casc, ii.copy(), cap, sf, 1.0, mx, step)
    ro = oracle.lbp_detect(casc, ii, cap, sf, 1.0, mx, step)
    assert_same(r, ro, "gs_lbp_detect %dx%d sf=%g max=%g step=%d cap=%d" % (w, h, sf, mx, step, cap))


def _body_resize_any_shape(emu, oracle, w, h, dw, dh, seed, nn):
    img = _img(np.random.RandomState(seed), w, h, seed % 3)
    d = np.full((dh, dw), 0xAB, np.uint8)
    emu.resize(d, img.copy(), nn)
    assert_same(d, oracle.resize(img, dw, dh, nn), "gs_resize%s %dx%d -> %dx%d" % ("_nn" if nn else "", w, h, dw, dh))
    if w >= 2 and h >= 2:
        d = np.full((h // 2, w // 2), 0xAB, np.uint8)
        emu.downsample(d, img.copy())
        assert_same(d, oracle.downsample(img), "gs_downsample %dx%d" % (w, h))


def _body_template_any_shape(emu, oracle, w, h, seed, data):
    rs = np.random.RandomState(seed)
    img = _img(rs, w, h, seed % 3)
    tw, th = data.draw(st.integers(1, w)), data.draw(st.integers(1, h))
    t = rs.randint(0, 256, (th, tw)).astype(np.uint8)
    r = np.full((h - th + 1, w - tw + 1), 0xAB, np.uint8)
    emu.match_template(img.copy(), t, r)
    ro = oracle.match_template(img, t)
    assert_same(r, ro, "gs_match_template %dx%d in %dx%d" % (tw, th, w, h))
    assert emu.find_best_match(r) == oracle.find_best_match(ro)

def _body_crop_copy_any(emu, oracle, w, h, seed, data):
    img = _img(np.random.RandomState(seed), w, h, seed % 3)
    rx, ry = data.draw(st.integers(0, w - 1)), data.draw(st.integers(0, h - 1))
    rw, rh = data.draw(st.integers(1, w - rx)), data.draw(st.integers(1, h - ry))
    d = np.full((rh, rw), 0xAB, np.uint8)
    emu.crop(d, img.copy(), rx, ry, rw, rh)
    assert_same(d, oracle.crop(img, rx, ry, rw, rh), "gs_crop (%d,%d,%d,%d) of %dx%d" % (rx, ry, rw, rh, w, h))
    d = np.full_like(img, 0xAB)
    emu.copy(d, img.copy())
    assert_same(d, img, "gs_copy %dx%d" % (w, h))


def _body_orb_drivers_any(emu, oracle, w, h, seed, nkps, threshold, levels, n):
    """the pyramid driver (nanomagick.c:245-290) and the same-size batch driver vs per-call oracle"""
    rs = np.random.RandomState(seed)
    mem = pc.Mem("host" if "kernel_emu" in emu.path else "device")
    img = _img(rs, w, h, seed % 3)
    pc.orb_pyramid(emu, oracle, img, mem, nkps=nkps, threshold=threshold, levels=levels, seed=seed)
    frames = np.stack([_img(rs, w, h, (seed + i) % 3) for i in range(n)])
    sm0 = rs.randint(0, 256, frames.shape).astype(np.uint8)
    got = emu.orb_extract_batch_dev(mem.put(frames), mem.put(sm0.copy()), nkps, threshold)
    for i in range(n):
        assert_same(got[i], oracle.orb_extract(frames[i], nkps, threshold, sm0[i]), "orb batch frame %d" % i)


# ---- the same properties through the emulator (CPU suite) and on the real GPU (-m gpu) ----------
@_cfg(40)
@given(w=widths, h=heights, seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2), radius=st.one_of(st.integers(0, 40), st.integers(41, 140)))
def test_stencils_any_shape(emu, oracle, w, h, seed, kind, radius):
    _body_stencils_any_shape(emu, oracle, w=w, h=h, seed=seed, kind=kind, radius=radius)


@pytest.mark.gpu
@_cfg(40)
@given(w=widths, h=heights, seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2), radius=st.one_of(st.integers(0, 40), st.integers(41, 140)))
def test_gpu_stencils_any_shape(hip, oracle, w, h, seed, kind, radius):
    _body_stencils_any_shape(hip, oracle, w=w, h=h, seed=seed, kind=kind, radius=radius)


@_cfg(25)
@given(w=widths, h=heights, seed=st.integers(0, 2 ** 16), norm=st.sampled_from([1, 2, 9, 16, 255, 256, 300]),
       ks=st.lists(st.integers(-16, 16), min_size=9, max_size=9))
def test_filter_any_shape(emu, oracle, w, h, seed, norm, ks):
    _body_filter_any_shape(emu, oracle, w=w, h=h, seed=seed, norm=norm, ks=ks)


@pytest.mark.gpu
@_cfg(25)
@given(w=widths, h=heights, seed=st.integers(0, 2 ** 16), norm=st.sampled_from([1, 2, 9, 16, 255, 256, 300]),
       ks=st.lists(st.integers(-16, 16), min_size=9, max_size=9))
def test_gpu_filter_any_shape(hip, oracle, w, h, seed, norm, ks):
    _body_filter_any_shape(hip, oracle, w=w, h=h, seed=seed, norm=norm, ks=ks)


@_cfg(20)
@given(n=st.integers(1, 5), w=st.sampled_from([32, 48, 64, 1040]), h=st.integers(3, 40), radius=st.integers(1, 3),
       seed=st.integers(0, 2 ** 16))
def test_fused_pipeline_any_shape(emu, oracle, n, w, h, radius, seed):
    _body_fused_pipeline_any_shape(emu, oracle, n=n, w=w, h=h, radius=radius, seed=seed)


@pytest.mark.gpu
@_cfg(20)
@given(n=st.integers(1, 5), w=st.sampled_from([32, 48, 64, 1040]), h=st.integers(3, 40), radius=st.integers(1, 3),
       seed=st.integers(0, 2 ** 16))
def test_gpu_fused_pipeline_any_shape(hip, oracle, n, w, h, radius, seed):
    _body_fused_pipeline_any_shape(hip, oracle, n=n, w=w, h=h, radius=radius, seed=seed)


@_cfg(20)
@given(w=st.integers(7, 90), h=st.integers(7, 60), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       threshold=st.sampled_from([0, 1, 5, 20, 60, 200, 255, 300]), nkps=st.integers(1, 80))
def test_fast_orb_match_any_shape(emu, oracle, w, h, seed, kind, threshold, nkps):
    _body_fast_orb_match_any_shape(emu, oracle, w=w, h=h, seed=seed, kind=kind, threshold=threshold, nkps=nkps)


@_cfg(12)
@given(w4=st.sampled_from([2, 3, 16, 63, 64, 65, 67, 128, 130]), h=st.integers(7, 30), seed=st.integers(0, 2 ** 16),
       kind=st.integers(0, 2), threshold=st.sampled_from([0, 1, 5, 20, 60, 200, 255, 256, 300]), dark=st.booleans())
def test_fast_px_kernel(emu, oracle, w4, h, seed, kind, threshold, dark):
    _body_fast_px_kernel(emu, oracle, w4=w4, h=h, seed=seed, kind=kind, threshold=threshold, dark=dark)


@pytest.mark.gpu
@_cfg(20)
@given(w4=st.sampled_from([2, 3, 16, 63, 64, 65, 67, 128, 130, 320]), h=st.integers(7, 90), seed=st.integers(0, 2 ** 16),
       kind=st.integers(0, 2), threshold=st.sampled_from([0, 1, 5, 20, 60, 200, 255, 256, 300]), dark=st.booleans())
def test_gpu_fast_px_kernel(hip, oracle, w4, h, seed, kind, threshold, dark):
    _body_fast_px_kernel(hip, oracle, w4=w4, h=h, seed=seed, kind=kind, threshold=threshold, dark=dark)


@_cfg(10)
@given(w=st.integers(60, 300), h=st.integers(40, 130), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       threshold=st.sampled_from([1, 12, 20, 60, 255]), cap=st.sampled_from([1, 50, 20000]), key19=st.sampled_from([0, 0, 1]))
def test_fast_wide_shapes(emu, oracle, w, h, seed, kind, threshold, cap, key19):
    _body_fast_wide(emu, oracle, w, h, seed, kind, threshold, cap, key19)


@pytest.mark.gpu
@_cfg(25)
@given(w=st.integers(60, 700), h=st.integers(40, 300), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       threshold=st.sampled_from([1, 12, 20, 60, 255]), cap=st.sampled_from([1, 50, 20000]), key19=st.sampled_from([0, 0, 1]))
def test_gpu_fast_wide_shapes(hip, oracle, w, h, seed, kind, threshold, cap, key19):
    _body_fast_wide(hip, oracle, w, h, seed, kind, threshold, cap, key19)


@pytest.mark.gpu
@_cfg(20)
@given(w=st.integers(7, 90), h=st.integers(7, 60), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       threshold=st.sampled_from([0, 1, 5, 20, 60, 200, 255, 300]), nkps=st.integers(1, 80))
def test_gpu_fast_orb_match_any_shape(hip, oracle, w, h, seed, kind, threshold, nkps):
    _body_fast_orb_match_any_shape(hip, oracle, w=w, h=h, seed=seed, kind=kind, threshold=threshold, nkps=nkps)


@_cfg(15)
@given(w=st.integers(24, 70), h=st.integers(24, 60), seed=st.integers(0, 2 ** 16), cseed=st.integers(0, 50),
       sf=st.sampled_from([1.1, 1.25, 1.5, 2.0]), mx=st.sampled_from([1.0, 1.6, 2.5]), step=st.integers(1, 3),
       cap=st.sampled_from([1, 7, 4096]))
def test_lbp_any_shape(emu, oracle, w, h, seed, cseed, sf, mx, step, cap):
    _body_lbp_any_shape(emu, oracle, w=w, h=h, seed=seed, cseed=cseed, sf=sf, mx=mx, step=step, cap=cap)


@pytest.mark.gpu
@_cfg(15)
@given(w=st.integers(24, 70), h=st.integers(24, 60), seed=st.integers(0, 2 ** 16), cseed=st.integers(0, 50),
       sf=st.sampled_from([1.1, 1.25, 1.5, 2.0]), mx=st.sampled_from([1.0, 1.6, 2.5]), step=st.integers(1, 3),
       cap=st.sampled_from([1, 7, 4096]))
def test_gpu_lbp_any_shape(hip, oracle, w, h, seed, cseed, sf, mx, step, cap):
    _body_lbp_any_shape(hip, oracle, w=w, h=h, seed=seed, cseed=cseed, sf=sf, mx=mx, step=step, cap=cap)


@_cfg(20)
@given(w=st.integers(1, 70), h=st.integers(1, 50), dw=st.integers(1, 90), dh=st.integers(1, 60),
       seed=st.integers(0, 2 ** 16), nn=st.booleans())
def test_resize_any_shape(emu, oracle, w, h, dw, dh, seed, nn):
    _body_resize_any_shape(emu, oracle, w=w, h=h, dw=dw, dh=dh, seed=seed, nn=nn)


@pytest.mark.gpu
@_cfg(20)
@given(w=st.integers(1, 70), h=st.integers(1, 50), dw=st.integers(1, 90), dh=st.integers(1, 60),
       seed=st.integers(0, 2 ** 16), nn=st.booleans())
def test_gpu_resize_any_shape(hip, oracle, w, h, dw, dh, seed, nn):
    _body_resize_any_shape(hip, oracle, w=w, h=h, dw=dw, dh=dh, seed=seed, nn=nn)


@_cfg(20)
@given(w=st.one_of(st.integers(1, 60), st.integers(61, 160)), h=st.one_of(st.integers(1, 40), st.integers(41, 70)), seed=st.integers(0, 2 ** 16), data=st.data())
def test_template_any_shape(emu, oracle, w, h, seed, data):
    _body_template_any_shape(emu, oracle, w=w, h=h, seed=seed, data=data)


@pytest.mark.gpu
@_cfg(20)
@given(w=st.one_of(st.integers(1, 60), st.integers(61, 160)), h=st.one_of(st.integers(1, 40), st.integers(41, 70)), seed=st.integers(0, 2 ** 16), data=st.data())
def test_gpu_template_any_shape(hip, oracle, w, h, seed, data):
    _body_template_any_shape(hip, oracle, w=w, h=h, seed=seed, data=data)

@_cfg(20)
@given(w=st.integers(1, 70), h=st.integers(1, 40), seed=st.integers(0, 2 ** 16), data=st.data())
def test_crop_copy_any(emu, oracle, w, h, seed, data):
    _body_crop_copy_any(emu, oracle, w, h, seed, data)


@pytest.mark.gpu
@_cfg(20)
@given(w=st.integers(1, 70), h=st.integers(1, 40), seed=st.integers(0, 2 ** 16), data=st.data())
def test_gpu_crop_copy_any(hip, oracle, w, h, seed, data):
    _body_crop_copy_any(hip, oracle, w, h, seed, data)


_orb_drv = dict(w=st.integers(16, 90), h=st.integers(16, 70), seed=st.integers(0, 2 ** 16), nkps=st.integers(1, 60),
                threshold=st.sampled_from([5, 20, 60]), levels=st.integers(1, 4), n=st.integers(1, 3))


@_cfg(10)
@given(**_orb_drv)
def test_orb_drivers_any(emu, oracle, w, h, seed, nkps, threshold, levels, n):
    _body_orb_drivers_any(emu, oracle, w, h, seed, nkps, threshold, levels, n)


@pytest.mark.gpu
@_cfg(10)
@given(**_orb_drv)
def test_gpu_orb_drivers_any(hip, oracle, w, h, seed, nkps, threshold, levels, n):
    _body_orb_drivers_any(hip, oracle, w, h, seed, nkps, threshold, levels, n)


@pytest.mark.gpu
@_cfg(25)
@given(n=st.integers(2, 70), per=st.integers(1, 70), w=st.sampled_from([32, 48, 64, 256]), h=st.integers(3, 30),
       radius=st.integers(1, 3), seed=st.integers(0, 2 ** 16))
def test_gpu_pipeline_chunking_any(hip, oracle, n, per, w, h, radius, seed):
    """gsh_edge_pipeline_batch with arbitrary chunk sizes of its side-stream overlap (gsh_tune key 5) gives
    the bytes of the unsplit call; a few frames are also checked against the oracle"""
    import torch
    rs = np.random.RandomState(seed)
    src = torch.from_numpy(np.stack([_img(rs, w, h, i % 3) for i in range(n)])).cuda()
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    ref, tref = torch.full_like(src, 1), torch.zeros(n, dtype=torch.uint8, device="cuda")
    out, thr = torch.full_like(src, 2), torch.zeros(n, dtype=torch.uint8, device="cuda")
    try:
        hip.tune(5, -1)
        hip.edge_pipeline_batch(ref, None, src, radius, hist, tref)
        href = hist.clone()
        hip.tune(5, per)
        for _ in range(2):  # twice: the side stream is rejoined and reused
            hist.zero_()
            hip.edge_pipeline_batch(out, None, src, radius, hist, thr)
        hip.sync()
        assert bool((out == ref).all()) and bool((thr == tref).all()) and bool((hist == href).all())
    finally:
        hip.tune(5, 0)
    for f in {0, n // 2, n - 1}:
        s = oracle.sobel(oracle.blur(src[f].cpu().numpy(), radius))
        t = oracle.otsu_threshold(s)
        assert int(tref[f]) == t
        assert_same(ref[f].cpu().numpy(), oracle.threshold(s, t), "frame %d" % f)


# ---- round 2: device-resident ORB (GS_NO_STDLIB trig) and score maps of another size --------------------
_CACHE = {}


def _body_orb_nostdlib_any_shape(g, w, h, seed, kind, threshold, nkps, nframes):
    from oracle import pyoracle
    from oracle.pyoracle import Oracle
    if not pyoracle.have_reference_nostdlib():
        pytest.skip("oracle/_ref/libgs_ref_nostdlib.so not present")
    rs = np.random.RandomState(seed)
    frames = np.stack([_img(rs, w, h, kind) for _ in range(nframes)])
    if "ref_ns" not in _CACHE:
        _CACHE["ref_ns"] = Oracle("reference_nostdlib")
    # gs_orb_extract needs w, h >= 31 for any keypoint to survive the 15-px border; smaller frames must give 0
    pc.orb_nostdlib(g, _CACHE["ref_ns"], frames, nkps=nkps, threshold=threshold)


@_cfg(12)
@given(w=st.integers(7, 96), h=st.integers(7, 72), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       threshold=st.sampled_from([1, 5, 20, 60, 200]), nkps=st.sampled_from([1, 3, 40, 1300]), nframes=st.integers(1, 3))
def test_orb_nostdlib_any_shape(emu, w, h, seed, kind, threshold, nkps, nframes):
    _body_orb_nostdlib_any_shape(emu, w=w, h=h, seed=seed, kind=kind, threshold=threshold, nkps=nkps, nframes=nframes)


@pytest.mark.gpu
@_cfg(12)
@given(w=st.integers(7, 96), h=st.integers(7, 72), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       threshold=st.sampled_from([1, 5, 20, 60, 200]), nkps=st.sampled_from([1, 3, 40, 1300]), nframes=st.integers(1, 3))
def test_gpu_orb_nostdlib_any_shape(hip, w, h, seed, kind, threshold, nkps, nframes):
    _body_orb_nostdlib_any_shape(hip, w=w, h=h, seed=seed, kind=kind, threshold=threshold, nkps=nkps, nframes=nframes)


def _body_fast_any_scoremap(g, reference, w, h, sw, sh, seed, threshold):
    rs = np.random.RandomState(seed)
    img = _img(rs, w, h, seed % 3)
    sm0 = rs.randint(0, 4, (sh, sw)).astype(np.uint8)
    sm = sm0.copy()
    got = g.fast(img, sm, 300, threshold)
    exp, sm_exp = reference.fast(img, 300, threshold, scoremap=sm0)
    assert_same(got, exp, "gs_fast %dx%d with a %dx%d score map" % (w, h, sw, sh))
    assert_same(sm, sm_exp, "score map after the call")


@_cfg(15)
@given(w=st.integers(7, 70), h=st.integers(7, 50), sw=st.integers(1, 80), sh=st.integers(1, 60), seed=st.integers(0, 2 ** 16),
       threshold=st.sampled_from([0, 5, 20, 100]))
def test_fast_any_scoremap_size(emu, reference, w, h, sw, sh, seed, threshold):
    _body_fast_any_scoremap(emu, reference, w=w, h=h, sw=sw, sh=sh, seed=seed, threshold=threshold)


# ---- histogram / Otsu of a batch: any frame size (every base alignment occurs inside a batch), any piece size of the
# ---- huge-image path (gsh_tune key 12), any blocks-per-frame choice (key 11)
def _body_histogram_batch(g, oracle, n, w, h, seed, kind, piece, bpf):
    rs = np.random.RandomState(seed)
    a = np.stack([_img(rs, w, h, kind) for _ in range(n)])
    hist = np.zeros((n, 256), np.uint32)
    thr = np.zeros(n, np.uint8)
    try:
        g.tune(12, piece)
        g.tune(11, bpf)
        g.histogram_batch(a, hist)
        for f in range(n):
            assert_same(hist[f], oracle.histogram(a[f]), "histogram frame %d of %d, %dx%d, piece %d, bpf %d" % (f, n, w, h, piece, bpf))
        g.otsu_batch(a, hist, thr)
        assert [int(t) for t in thr] == [oracle.otsu_threshold(a[f]) for f in range(n)]
    finally:
        g.tune(12, 0)
        g.tune(11, 0)


@_cfg(30)
@given(n=st.integers(1, 4), w=st.integers(1, 300), h=st.integers(1, 40), seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2),
       piece=st.sampled_from([0, 0, 17, 100, 999, 4096]), bpf=st.sampled_from([0, 0, 1, 2, 5]))
def test_histogram_batch_any_shape(emu, oracle, n, w, h, seed, kind, piece, bpf):
    _body_histogram_batch(emu, oracle, n, w, h, seed, kind, piece, bpf)
