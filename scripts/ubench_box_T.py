import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
W, H = 3840, 2160
def timeit(fn, reps=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for F in (64, 8):
    src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    dst = torch.zeros_like(src)
    for name, fn in (("blur r=5", lambda: g.blur_batch(dst, src, 5)), ("blur r=16", lambda: g.blur_batch(dst, src, 16)), ("blur r=40", lambda: g.blur_batch(dst, src, 40)), ("blur r=100", lambda: g.blur_batch(dst, src, 100)),
                     ("adaptive r=8", lambda: g.adaptive_threshold_batch(dst, src, 8, 5)), ("adaptive r=25", lambda: g.adaptive_threshold_batch(dst, src, 25, 5))):
        row = []
        for T in (0, 17, 34, 68, 135):
            g.tune(0, T); row.append("%d:%.3f" % (T, timeit(fn)))
        g.tune(0, 0)
        print("%2d frames %-14s ms by rows per band  %s" % (F, name, "  ".join(row)))
