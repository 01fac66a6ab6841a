#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache serve reads of data a kernel has just WRITTEN?  gs_threshold (in place: 1 R + 1 W per px) on n 4K frames
right after gs_blur wrote them ("warm") vs after 600 MB of other traffic ("cold").  If warm were much faster, a threshold pass
running a few frames behind the fused blur+sobel kernel could skip its HBM reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
W, H = 3840, 2160
big = torch.empty((72, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(big, 7)
big2 = torch.empty_like(big)
def t_ms(fn_pre, fn, reps=8):
    tot = 0.0
    for _ in range(reps):
        fn_pre(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    return tot / reps
for n in (2, 4, 8, 16, 32):
    src = big[:n]; dst = torch.empty_like(src)
    warm = t_ms(lambda: g.blur_batch(dst, src, 2), lambda: g.threshold_batch(dst, 100))
    cold = t_ms(lambda: (g.blur_batch(dst, src, 2), g.blur_batch(big2, big, 2)), lambda: g.threshold_batch(dst, 100))
    warm_s = t_ms(lambda: g.sobel_batch(dst, src), lambda: g.threshold_batch(dst, 100))
    mb = n * W * H / 1e6
    print("%2d frames (%4.0f MB): threshold right after blur wrote it %.4f ms (%.2f TB/s R+W), after sobel %.4f ms, after 600 MB of other traffic %.4f ms (%.2f TB/s)" % (
        n, mb, warm, 2 * mb / warm / 1e3, warm_s, cold, 2 * mb / cold / 1e3))
