#!/usr/bin/env python3
"""gs_histogram per call (k_hist_partial + k_hist_reduce) and gs_histogram + gs_otsu_threshold: pixels per second over
512 x 3840x2160 frames (4.2 GB, read once) and smaller batches; gsh_tune key 10 = trips per block the launcher aims at (default 48), key 11 = blocks per frame"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("UB_TAG", "")
g = gs.Grayskull(os.path.join(ROOT, "build_variants", "libgs_%s.so" % TAG)) if TAG else gs.lib()
g.use_torch_stream()
print("# library:", TAG or "this tree (depth 3)")
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (n, h, w) in [(512, 2160, 3840), (64, 4096, 4096), (32, 720, 1280), (1, 2160, 3840), (8, 1080, 1920)]:
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda"); thr = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ref = torch.stack([torch.bincount(src[i].flatten().to(torch.int64), minlength=256) for i in (0, n - 1)]).to(torch.int32)
    for bpf in [int(x) for x in os.environ.get("UB_TRIPS", "0").split(",")]:  # >= 1000: 1000 * (threads per block / 256) + trips
        g.tune(10, bpf)
        ms = timeit(lambda: g.histogram_batch(src, hist))
        ok = bool((hist[[0, n - 1]] == ref).all())
        ms2 = timeit(lambda: g.otsu_batch(src, hist, thr))
        print("%4d x %dx%d  trips %3d  histogram %.4f ms = %.2f Tpx/s (%.3f of 8 TB/s)   +otsu %.4f ms = %.2f Tpx/s   exact=%s" % (
            n, w, h, bpf, ms, n * h * w / ms / 1e9, n * h * w / ms / 1e9 / 8.0, ms2, n * h * w / ms2 / 1e9, ok))
    g.tune(10, 0)
    del src
