#!/usr/bin/env python3
"""what the dense phase of k_lbp_tile is sensitive to: the release kernels (build_variants/libgs_experiment.so) against builds with
12 more corner reads (libgs_sens1.so) or ~16 more VALU operations (libgs_sens2.so) per dense classifier evaluation -- 8 x 4K edge
maps, 8 x 4K block noise, 8 x 1080p block noise; one process per library (UB_LIB), run alternately"""
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
out = []
for (kind, w, h, n) in (("edges", 3840, 2160, 8), ("noise", 3840, 2160, 8), ("noise", 1920, 1080, 8)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    if kind == "edges":
        a, b = torch.empty_like(src), torch.zeros_like(src)
        g.blur_batch(a, src, 2); g.sobel_batch(b, a); src = b
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
    rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    g.tune(13, 2)  # eighths: no max_rects effect on the comparison
    ms = min(timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)) for _ in range(2))
    out.append("%s %dx%d %.3f" % (kind, w, h, ms / n))
print("%-22s %s" % (os.path.basename(os.environ.get("UB_LIB", "in-tree")), "  ".join(out)), flush=True)
