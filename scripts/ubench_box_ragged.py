#!/usr/bin/env python3
"""sliding box on a ragged and an aligned batch (64 frames): gs_blur r = 5, 9, 16 and gs_adaptive_threshold r = 15, wall time per
call by stream events; ragged rows with k_box_edge on the side stream (default), on the caller's stream (gsh_tune key 6 = 6) and on
the any-radius kernel (key 6 = 5, rounds 2-4)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (w, h, n) in ((3840, 2160, 64), (3838, 2160, 64), (1920, 1080, 64), (1918, 1080, 64), (1080, 1920, 64), (1366, 768, 256)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 5)
    dst = torch.zeros_like(src)
    for rnd in range(2):
        for key in ((0,) if w % 16 == 0 else (0, 6, 5)):
            g.tune(6, key)
            t = [timeit(lambda: g.blur_batch(dst, src, r)) for r in (5, 9, 16)] + [timeit(lambda: g.adaptive_threshold_batch(dst, src, 15, 5))]
            print("%4d x %4d x %3d  key 6 = %d  blur r=5 %.4f  r=9 %.4f  r=16 %.4f  adaptive r=15 %.4f ms" % (w, h, n, key, *t), flush=True)
    g.tune(6, 0)
