"""shared helpers for the test-suite"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_pgm(path):
    raw = open(path, "rb").read()
    parts = raw.split(b"\n", 3)
    assert parts[0] == b"P5"
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8, w * h).reshape(h, w).copy()


def lena():
    return read_pgm(os.path.join(GOLDEN, "lena.pgm"))


def fnv(a):
    from oracle.pyoracle import Oracle
    return "%08x" % Oracle.fnv1a(a)


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return "shape %s vs %s" % (a.shape, b.shape)
    if a.dtype.names:
        a, b = a.view(np.uint8), b.view(np.uint8)
    d = np.argwhere(a != b)
    if len(d) == 0:
        return None
    i = tuple(d[0])
    return "%d mismatches, first at %s: got %s expected %s" % (len(d), i, a[i], b[i])


def assert_same(got, exp, what=""):
    msg = first_diff(got, exp)
    assert msg is None, "%s: %s" % (what, msg)


def random_cascade(seed, nstages=3, weaks_per_stage=3, window=24, permissive=True):
    """a small random LBP cascade that lets many windows through (exercises ordering/caps)"""
    from grayskull_amd.cascade import Cascade
    rs = np.random.RandomState(seed)
    nw = nstages * weaks_per_stage
    nf = nw + 2
    feats = np.zeros((nf, 4), np.int8)
    for i in range(nf):
        fw, fh = rs.randint(1, 5), rs.randint(1, 5)
        feats[i] = (rs.randint(0, window - 3 * fw + 1), rs.randint(0, window - 3 * fh + 1), fw, fh)
    left = rs.uniform(-1, 1, nw).astype(np.float32)
    right = rs.uniform(-1, 1, nw).astype(np.float32)
    thr = np.full(nstages, -0.6 if permissive else 0.3, np.float32)
    return Cascade(window, window, features=feats.reshape(-1),
                   weak_feature_idx=rs.randint(0, nf, nw).astype(np.uint16),
                   weak_left_val=left, weak_right_val=right,
                   weak_subset_offset=(np.arange(nw) * 8).astype(np.uint16),
                   weak_num_subsets=np.full(nw, 8, np.uint16),
                   subsets=rs.randint(-2**31, 2**31 - 1, nw * 8).astype(np.int32),
                   stage_weak_start=(np.arange(nstages) * weaks_per_stage).astype(np.uint16),
                   stage_nweaks=np.full(nstages, weaks_per_stage, np.uint16),
                   stage_threshold=thr)
