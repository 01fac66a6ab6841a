#!/usr/bin/env python3
"""LBP cascade: XCD-aware chunk mapping (default) vs chunks in dispatch order (gsh_tune key 13 = 1) on block-noise frames (configs[2])
and edge maps (configs[4]), 1080p and 4K; identical rectangles checked by checksum"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.lib(); g.use_torch_stream()
dc = g.cascade_create(Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin")))
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (w, h, n) in ((1920, 1080, 8), (3840, 2160, 4), (1280, 720, 8)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    a, b = torch.empty_like(src), torch.zeros_like(src)
    g.blur_batch(a, src, 2); g.sobel_batch(b, a)
    for name, img in (("noise", src), ("edges", b)):
        ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(img, ii)
        rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
        for rep in range(2):
            for key13 in (1, 0):
                g.tune(13, key13)
                ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
                crc = zlib.crc32(rects.cpu().numpy().tobytes()) ^ zlib.crc32(counts.cpu().numpy().tobytes())
                print("%s %dx%d  %-22s %.3f ms/frame  counts %s  crc %08x" % (name, w, h, "dispatch order" if key13 else "XCD-aware (default)", ms / n, counts.cpu().tolist()[:2], crc))
        g.tune(13, 0)
dc.close()
