#!/usr/bin/env python3
"""throughput of the SURVEY 8(f) "next" rows on 64 x 3840x2160 frames: gs_adaptive_threshold (radius 2 / 8 / 25), gs_filter 3x3,
gs_downsample, gs_blur radius 5 / 16 (sliding box), next to gs_blur(2) and the strip copy.  GB/s = algorithmic bytes / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()  # UB_LIB=build_variants/libgs_experiment.so for the strip-copy probe
W, H, F = 3840, 2160, 64
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
half = torch.zeros((F, H // 2, W // 2), dtype=torch.uint8, device="cuda")
npx = F * W * H
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
kern = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.int32)
ops = [("strip copy", lambda: g.probe_strip_copy(dst, src), 2.0),
       ("gs_blur r=2", lambda: g.blur_batch(dst, src, 2), 2.0),
       ("gs_blur r=5 (box)", lambda: g.blur_batch(dst, src, 5), 2.0),
       ("gs_blur r=16 (box)", lambda: g.blur_batch(dst, src, 16), 2.0),
       ("gs_adaptive_threshold r=2", lambda: g.adaptive_threshold_batch(dst, src, 2, 5), 2.0),
       ("gs_adaptive_threshold r=8", lambda: g.adaptive_threshold_batch(dst, src, 8, 5), 2.0),
       ("gs_adaptive_threshold r=25", lambda: g.adaptive_threshold_batch(dst, src, 25, 5), 2.0),
       ("gs_filter 3x3 norm 16", lambda: g.filter_batch(dst, src, kern, 16), 2.0),
       ("gs_downsample", lambda: g.downsample_batch(half, src), 1.25)]
for name, fn, bpp in ops:
    ms = timeit(fn)
    print("%-28s %.4f ms  %8.0f Mpix/s  %6.1f GB/s algorithmic (%.2f of 8 TB/s)" % (name, ms, npx / ms / 1e3, bpp * npx / ms / 1e6, bpp * npx / ms / 1e6 / 8000))
