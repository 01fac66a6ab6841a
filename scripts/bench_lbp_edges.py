#!/usr/bin/env python3
"""LBP cascade on the frames of BASELINE configs[4] (edge maps: gs_sobel(gs_blur(synth, 2)), 3840x2160): survivor
re-packing presets (gsh_tune key 4, see launch_lbp_padded)"""
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
ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(b, ii)
dc = g.cascade_create(casc)
rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
ref = None
for rnd in range(2):
    for preset in (0, 2, 3, 4, 5, 6, 7):
        g.tune(4, preset)
        ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
        c = counts.cpu().tolist()
        if ref is None: ref = (c, rects.clone())
        print("edge maps 4K preset %d: %.3f ms/frame  same=%s" % (preset, ms / n, c == ref[0] and bool((rects == ref[1]).all())))
g.tune(4, 0); dc.close()
