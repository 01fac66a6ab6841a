"""Property test (hypothesis): for random small shapes -- biased towards the strip-kernel boundaries
(w a multiple of 16, tiny heights, radius >= height) -- every drop-in function run through the
kernel sources (host-fiber emulator) equals the oracle.  Complements the fixed shape lists."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import parity_cases as pc
from util import assert_same

MEM = pc.Mem("host")
widths = st.one_of(st.sampled_from([16, 32, 48, 64, 80, 96, 112, 1024, 1040]), st.integers(1, 70))
heights = st.one_of(st.integers(1, 12), st.integers(13, 70))


def _img(rs, w, h, kind):
    if kind == 0:
        return rs.randint(0, 256, (h, w)).astype(np.uint8)
    if kind == 1:  # flat blocks + small noise: many equal neighbours, clamps rarely hit
        return np.clip(rs.randint(0, 256) + rs.randint(-3, 4, (h, w)), 0, 255).astype(np.uint8)
    return (rs.randint(0, 2, (h, w)) * 255).astype(np.uint8)  # binary: saturating sums


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(w=widths, h=heights, seed=st.integers(0, 2 ** 16), kind=st.integers(0, 2), radius=st.integers(0, 40))
def test_stencils_any_shape(emu, oracle, w, h, seed, kind, radius):
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


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(w=widths, h=heights, seed=st.integers(0, 2 ** 16), norm=st.sampled_from([1, 2, 9, 16, 255, 256, 300]),
       ks=st.lists(st.integers(-16, 16), min_size=9, max_size=9))
def test_filter_any_shape(emu, oracle, w, h, seed, norm, ks):
    if w >= 1024 and h > 8:
        h = 8
    img = _img(np.random.RandomState(seed), w, h, seed % 3)
    k = np.array(ks, np.int8).reshape(3, 3)
    d = np.full_like(img, 0xAB)
    emu.filter(d, img.copy(), k, norm)
    assert_same(d, oracle.filter(img, k, norm), "gs_filter %dx%d norm=%d k=%s" % (w, h, norm, ks))


@settings(max_examples=20, deadline=None, suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 5), w=st.sampled_from([32, 48, 64, 1040]), h=st.integers(3, 40), radius=st.integers(1, 3),
       seed=st.integers(0, 2 ** 16))
def test_fused_pipeline_any_shape(emu, oracle, n, w, h, radius, seed):
    if w >= 1024 and h > 8:
        h = 8
    rs = np.random.RandomState(seed)
    src = np.stack([_img(rs, w, h, i % 3) for i in range(n)])
    out, hist, thr = np.full_like(src, 7), np.zeros((n, 256), np.uint32), np.zeros(n, np.uint8)
    emu.edge_pipeline_batch(out, None, src, radius, hist, thr)
    for i in range(n):
        s = oracle.sobel(oracle.blur(src[i], radius))
        t = oracle.otsu_threshold(s)
        assert int(thr[i]) == t, "otsu frame %d" % i
        assert_same(out[i], oracle.threshold(s, t), "pipeline frame %d (%dx%d r=%d)" % (i, w, h, radius))
