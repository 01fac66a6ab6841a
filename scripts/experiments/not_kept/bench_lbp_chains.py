#!/usr/bin/env python3
"""k_lbp_tile's stage 0 by chains across the block (default) against one window per lane in the waves' own phases
(gsh_tune key 15 = 1 + 16 * 7 + 256), with the tiles in eighths (key 13 = 2: no max_rects effect) and in scan order (0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
dc = g.cascade_create(Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin")))
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (kind, w, h, n) in (("edges", 3840, 2160, 8), ("noise", 3840, 2160, 8), ("noise", 1920, 1080, 8), ("edges", 1920, 1080, 8), ("noise", 1280, 720, 8)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    if kind == "edges":
        a, b = torch.empty_like(src), torch.zeros_like(src)
        g.blur_batch(a, src, 2); g.sobel_batch(b, a); src = b
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
    rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    ref = None
    for rnd in range(2):
        for k13 in (2, 0):
            for k15 in (0, 1 + 16 * 7 + 256):
                g.tune(13, k13); g.tune(15, k15)
                ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
                sig = (counts.cpu().tolist(), int(rects.sum()))
                ref = ref or sig
                print("%s %dx%d  tiles %-10s stage 0 %-16s %.3f ms/frame  same=%s" % (kind, w, h, "eighths" if k13 else "scan order", "by chains" if k15 == 0 else "window per lane", ms / n, sig == ref), flush=True)
    g.tune(13, 0); g.tune(15, 0)
