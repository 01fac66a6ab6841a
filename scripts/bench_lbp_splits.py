#!/usr/bin/env python3
"""LBP cascade: phase split points of the survivor re-packing (gsh_tune key 4 = 1000 + e0 + 32 e1 + 1024 e2 + 32768 e3)
on block-noise frames (configs[2]) and on edge maps (configs[4])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.lib(); g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
w, h, n = 3840, 2160, 4
src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
a, b = torch.empty_like(src), torch.zeros_like(src)
g.blur_batch(a, src, 2); g.sobel_batch(b, a)
ii_e = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(b, ii_e)
ii_n = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii_n)
dc = g.cascade_create(casc)
rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
splits = [(2, 4, 7), (3, 6, 10), (4, 8), (3, 5, 8), (4, 6, 9)]
for name, ii in (("noise", ii_n), ("edges", ii_e)):
    ref = None
    for sp in [("adaptive", 6, 2), ("adaptive", 8, 2), ("adaptive", 10, 2), ("adaptive", 12, 2), ("adaptive", 15, 2), ("adaptive", 10, 2, 2, 4, 0), ("adaptive", 10, 2, 1, 3, 6)] + splits[:1]:
        if sp[0] == "adaptive": g.tune(4, 0); g.tune(9, sp[1] + 16 * sp[2] + (256 * sp[3] + 4096 * sp[4] + 65536 * sp[5] if len(sp) > 3 else 0))
        else: g.tune(4, 1000 + sum(e << (5 * i) for i, e in enumerate(sp)))
        ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
        c = counts.cpu().tolist()
        if ref is None: ref = (c, rects.clone())
        print("%s 4K splits %-12s %.3f ms/frame  same=%s" % (name, sp, ms / n, c == ref[0] and bool((rects == ref[1]).all())))
g.tune(4, 0); g.tune(9, 0); dc.close()
