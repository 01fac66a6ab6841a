#!/usr/bin/env python3
"""LBP cascade: sweep of the survivor-compaction split points (gsh_tune 4/5) on 1080p and 4K"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (w, h, n, seed) in ((1920, 1080, 8, 3), (3840, 2160, 4, 1000)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, seed)
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
    dc = g.cascade_create(casc)
    rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    nwin = g.lbp_window_count(casc, w, h, 1.1, 1.0, 4.0, 1)
    ref = None
    for (s1, s2) in ((1, 0), (0, 0), (2, 0), (4, 0), (5, 0)):
        g.tune(4, s1)
        ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
        c = counts.cpu().tolist()
        if ref is None: ref = (c, rects.clone())
        ok = c == ref[0] and bool((rects == ref[1]).all())
        print("%dx%d preset=(%d,%d) ms/frame=%.3f Gwin/s=%.2f same=%s" % (w, h, s1, s2, ms / n, nwin * n / ms / 1e6, ok))
    dc.close()
g.tune(4, 0); g.tune(5, 0)
