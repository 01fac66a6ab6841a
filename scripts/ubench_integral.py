#!/usr/bin/env python3
"""gsh_integral_batch: ms per frame vs frames per call (does a group that fits the 256 MiB Infinity Cache
re-read its source from there?) and vs calling it in groups"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
W, H = 3840, 2160
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
F = 64
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
ii = torch.empty((F, H, W), dtype=torch.int32, device="cuda")
for n in (4, 8, 12, 16, 24, 32, 64):
    def run():
        for f0 in range(0, F, n): g.integral_batch(src[f0:f0 + n], ii[f0:f0 + n])
    ms = timeit(run)
    print("groups of %2d frames: %.4f ms per frame, %.0f GB/s (5 B/px)" % (n, ms / F, 5.0 * F * W * H / ms / 1e6))
