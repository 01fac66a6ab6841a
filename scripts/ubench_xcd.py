#!/usr/bin/env python3
"""strip kernels: bands in dispatch order (gsh_tune key 18 = 1) vs the XCD-aware band mapping (key 18 = 2) vs the
default rule, interleaved rounds in ONE process; median GB/s of algorithmic 2 B/px.  UB_W / UB_H / UB_F select the batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib()
g.use_torch_stream()
W, H, F = int(os.environ.get("UB_W", 3840)), int(os.environ.get("UB_H", 2160)), int(os.environ.get("UB_F", 64))
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
npx = F * W * H
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
kern = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], np.int8)
ops = {"copy": lambda: g.probe_strip_copy(dst, src), "erode": lambda: g.erode_batch(dst, src),
       "sobel": lambda: g.sobel_batch(dst, src), "blur2": lambda: g.blur_batch(dst, src, 2),
       "blur1": lambda: g.blur_batch(dst, src, 1), "filter": lambda: g.filter_batch(dst, src, kern, 1)}
res = {}
Ts = [int(x) for x in os.environ.get("UB_T", "0").split(",")]
for rnd in range(3):
    for op in os.environ.get("UB_OPS", "copy,erode,sobel,blur2,filter").split(","):
        for T in Ts:
            for k18 in (1, 2, 0):
                g.tune(0, T); g.tune(18, k18)
                res.setdefault((op, T, k18), []).append(timeit(ops[op], 5))
g.tune(0, 0); g.tune(18, 0)
names = {1: "dispatch order", 2: "XCD-aware", 0: "default rule"}
for (op, T, k18), v in sorted(res.items()):
    ms = float(np.median(v)); gbs = 2.0 * npx / ms / 1e6
    print("%dx%d x%d  %-6s T %3d  %-15s %8.4f ms  %7.1f GB/s  frac %.3f" % (W, H, F, op, T, names[k18], ms, gbs, gbs / 8000), flush=True)
