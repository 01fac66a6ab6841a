#!/usr/bin/env python3
"""config-2 pipeline: batch size x chunking of the side-stream overlap (gsh_tune key 5:
-1 never split, N fixed N-frame chunks, 0 default = 32-frame chunks with a tapered tail)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("UB_TAG", "")
g = gs.Grayskull(os.path.join(ROOT, "build_variants", "libgs_%s.so" % TAG)) if TAG else gs.lib()
g.use_torch_stream()
print("# library:", TAG or "this tree")
W, H = 3840, 2160
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for F in [int(x) for x in os.environ.get("UB_F", "64,128,256,512").split(",")]:
    src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    dst = torch.zeros_like(src)
    hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda"); thr = torch.zeros((F,), dtype=torch.uint8, device="cuda")
    ref = None
    for rnd in range(2):
        for per in [int(x) for x in os.environ.get("UB_PER", "-1,32,64,0").split(",")]:
            if per > 0 and per >= F: continue
            g.tune(5, per)
            ms = timeit(lambda: g.edge_pipeline_batch(dst, None, src, 2, hist, thr))
            cs = int(dst.view(torch.int32).sum().item()) & 0xffffffff
            if ref is None: ref = cs
            print("frames %4d chunking %4d  %.4f ms  %.0f Mpix/s  %s" % (F, per, ms, F * W * H / ms / 1e3, "ok" if cs == ref else "MISMATCH"))
    del src, dst
g.tune(5, 0)
