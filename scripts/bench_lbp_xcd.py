#!/usr/bin/env python3
"""chunk / tile -> XCD mapping of the LBP kernels (gsh_tune key 13: 0 = the rule, 1 = dispatch order, 2 = eighths always, 3 = tile rows dealt
round the XCDs [k_lbp_cascade: eighths], 4 = tiles in dispatch order [k_lbp_cascade: eighths]) with the rule's kernels: 8 frames of
edge maps and of block noise at 4K, 1440p, 1080p, 720p.  (First version of this script, profiles/r05j_lbp_xcd.log: keys 0 1 2 with
the rule and with k_lbp_cascade for every scale.)"""
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
for (kind, w, h, n) in (("edges", 3840, 2160, 8), ("noise", 3840, 2160, 8), ("edges", 2560, 1440, 8), ("noise", 2560, 1440, 8),
                        ("edges", 1920, 1080, 8), ("noise", 1920, 1080, 8), ("edges", 1280, 720, 8), ("noise", 1280, 720, 8)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    if kind == "edges":
        a, b = torch.empty_like(src), torch.zeros_like(src)
        g.blur_batch(a, src, 2); g.sobel_batch(b, a); src = b
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
    rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    for rnd in range(2):
        for k14 in (0,):
            for k13 in (0, 2):
                g.tune(14, k14); g.tune(13, k13)
                ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
                print("%s %dx%d %-14s key 13 = %d: %.3f ms/frame" % (kind, w, h, "rule" if k14 == 0 else "k_lbp_cascade", k13, ms / n), flush=True)
    g.tune(14, 0); g.tune(13, 0)
    if os.environ.get("PER_SCALE") and kind == "edges" and w in (3840, 2560):
        s = 1.0
        while s <= 4.0:
            line = "%s %dx%d scale %.3f:" % (kind, w, h, s)
            for k13 in (2, 4, 2, 4):
                g.tune(13, k13)
                ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, s, s * 1.05, 1))
                line += "  key 13 = %d %.4f" % (k13, ms / n)
            print(line, flush=True)
            s = float(torch.tensor(s, dtype=torch.float32) * torch.tensor(1.1, dtype=torch.float32))
        g.tune(13, 0)
dc.close()
