#!/usr/bin/env python3
"""LBP cascade (quad-lane survivors): sweep of the adaptive re-packing rule (gsh_tune key 9 = max stages + 16 * tenths
+ 256 * d1 + 4096 * d2 + 65536 * d3) on 8 x 4K block-noise frames and edge maps; identical rectangles checked."""
import os, sys, zlib
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
w, h, n = int(os.environ.get("UB_W", 3840)), int(os.environ.get("UB_H", 2160)), int(os.environ.get("UB_F", 8))
src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
a, b = torch.empty_like(src), torch.zeros_like(src)
g.blur_batch(a, src, 2); g.sobel_batch(b, a)
rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
g.tune(14, 0)
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1].split(";")]
else:
    combos = [(8, 2, 2, 5, 0)] + [(8, t, 2, 5, 0) for t in (1, 3, 4, 5, 7)] + [(m, 3, 2, 5, 0) for m in (4, 6, 12)] + \
             [(8, 3, d1, d2, d3) for (d1, d2, d3) in ((1, 3, 6), (1, 2, 4), (2, 4, 8), (3, 6, 0), (1, 2, 3), (2, 0, 0), (1, 3, 0))]
for name, img in (("noise", src), ("edges", b)):
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(img, ii)
    ref = None
    for (m, t, d1, d2, d3) in combos:
        g.tune(9, m + 16 * t + 256 * d1 + 4096 * d2 + 65536 * d3)
        ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)) / n
        crc = zlib.crc32(rects.cpu().numpy().tobytes()) ^ zlib.crc32(counts.cpu().numpy().tobytes())
        ref = ref or crc
        print("%s %dx%d x%d adaptive max %2d tenths %d next +%d +%d +%d  %.3f ms/frame  same=%s" % (name, w, h, n, m, t, d1, d2, d3, ms, crc == ref), flush=True)
    g.tune(9, 0)
g.tune(14, 0); dc.close()
