#!/usr/bin/env python3
"""A/B of library variants (AB_LIBS = comma-separated paths; "base" = the in-tree library) on gs_lbp_detect:
8 x 1080p block noise (configs[2]) and 8 x 4K edge maps (configs[4]: blur 2 -> sobel -> integral)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = os.environ.get("AB_LIBS", "base").split(",")
libs = {t: (gs.lib() if t == "base" else gs.Grayskull(os.path.join(ROOT, t))) for t in tags}
for g in libs.values(): g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g0 = libs[tags[0]]
for (w, h, n, seed, edge) in ((1920, 1080, 8, 3, False), (3840, 2160, 8, 1000, True)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g0.synth_batch(src, seed)
    if edge:
        a = torch.zeros_like(src); b = torch.zeros_like(src)
        g0.blur_batch(a, src, 2); g0.sobel_batch(b, a); src = b
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g0.integral_batch(src, ii)
    ref = None
    for rnd in range(3):
        for t in tags:
            g = libs[t]
            dc = g.cascade_create(casc)
            rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
            ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
            sig = (counts.cpu().tolist(), int(rects.sum()))
            ref = ref or sig
            print("%dx%d %s  %-40s %.3f ms/frame  same=%s" % (w, h, "edge maps" if edge else "block noise", t, ms / n, sig == ref), flush=True)
            dc.close()
