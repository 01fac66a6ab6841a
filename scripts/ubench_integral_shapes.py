#!/usr/bin/env python3
"""gsh_integral_batch over the frame shapes the reviews quote (narrow fixtures of the reference in large batches, video frames
in small ones, 4K, 8K): ms per call and fraction of 8 TB/s at the algorithmic 5 B/px.  Under rocprofv3 --kernel-trace the
three launches (colsum / colbase / wave) show their split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
SHAPES = [(612, 816, 256), (608, 816, 256), (612, 816, 64), (640, 480, 256), (128, 128, 1024), (1280, 720, 32), (1280, 720, 256),
          (1920, 1080, 8), (1920, 1080, 64), (3840, 2160, 8), (3840, 2160, 64), (4096, 4096, 16), (7680, 4320, 8)]
for (w, h, n) in SHAPES:
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 7)
    ii = torch.empty((n, h, w), dtype=torch.int32, device="cuda")
    ms = timeit(lambda: g.integral_batch(src, ii))
    by = 5.0 * n * w * h
    print("gs_integral %5d x %-5d x %-4d  %8.4f ms  %7.1f GB/s  %.3f of 8 TB/s" % (w, h, n, ms, by / ms / 1e6, by / ms / 1e6 / 8000.0), flush=True)
    del src, ii
