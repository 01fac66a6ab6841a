#!/usr/bin/env python3
"""k_lbp_tile (corners from an LDS tile, per-wave phases, pair-parallel survivors) against k_lbp_cascade on the frames of
BASELINE configs[4] (4K edge maps) and configs[2] (1080p block noise): whole scans per tile shape (gsh_tune key 14), the rule's
choice under LDS ceilings (key 15), and one scale at a time so that the per-scale winner is visible.  Every variant's
rectangles are compared with k_lbp_cascade's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
dc = g.cascade_create(casc)
SHAPES = {1: "k_lbp_cascade", 2: "tile 512thr 128x32", 3: "tile 1024thr 128x32", 4: "tile 1024thr 64x32", 5: "tile 1024thr 64x16", 6: "tile 512thr 64x32", 0: "rule"}
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def frames(kind, w, h, n):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    if kind == "edges":
        a, b = torch.empty_like(src), torch.zeros_like(src)
        g.blur_batch(a, src, 2); g.sobel_batch(b, a); src = b
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
    return ii
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for (kind, w, h, n) in (("edges", 3840, 2160, 8), ("noise", 1920, 1080, 8), ("noise", 3840, 2160, 8), ("edges", 1920, 1080, 8)):
    ii = frames(kind, w, h, n)
    rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    run = lambda mn=1.0, mx=4.0: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, mn, mx, 1)
    ref = {}
    def check(key):
        c = counts.cpu().tolist()
        if key not in ref: ref[key] = (c, rects.clone())
        return c == ref[key][0] and bool((rects == ref[key][1]).all())
    for rnd in range(2):
        for mode in (1, 2, 3, 4, 5, 6, 0):
            g.tune(14, mode)
            ms = timeit(run)
            print("%s %dx%d whole scan, %-22s %.3f ms/frame  same=%s" % (kind, w, h, SHAPES[mode] + ":", ms / n, check("all")), flush=True)
    g.tune(14, 0)
    for first in (1, 2):
        for tenths in (4, 5, 6, 7, 8):  # a wave goes from dense to pair-parallel once <= tenths/10 of its windows are alive
            g.tune(15, first + 16 * tenths)
            ms = timeit(run)
            print("%s %dx%d whole scan, rule, dense stages >= %d, re-pack at <= %d/10 alive: %.3f ms/frame  same=%s" % (kind, w, h, first, tenths, ms / n, check("all")), flush=True)
    g.tune(15, 0)
    if quick or (w, h) != (3840, 2160): continue
    s = 1.0
    while s <= 4.0:
        line = "%s %dx%d scale %.3f:" % (kind, w, h, s)
        for mode in (1, 2, 3, 4, 5, 6):
            g.tune(14, mode)
            ms = timeit(lambda: run(s, s * 1.05))
            line += "  %s %.3f%s" % (SHAPES[mode].replace("tile ", "").replace("k_lbp_", ""), ms / n, "" if check("s%.3f" % s) else " DIFF")
        print(line, flush=True)
        s = float(torch.tensor(s, dtype=torch.float32) * torch.tensor(1.1, dtype=torch.float32))
    g.tune(14, 0)
dc.close()
