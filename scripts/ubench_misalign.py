#!/usr/bin/env python3
"""round 4: what a byte phase costs the strip kernels (run on the GPU box): 64 x 4K at base + off for off in 0..16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib()
g.use_torch_stream()
if os.environ.get("UB_TUNE24"): g.tune(24, int(os.environ["UB_TUNE24"]))
W, H, F = 3840, 2160, 64
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
sb = torch.randint(0, 256, (F * H * W + 64,), dtype=torch.uint8, device="cuda")
db = torch.zeros(F * H * W + 64, dtype=torch.uint8, device="cuda")
k3 = np.array([[1, -2, 1], [2, 4, -2], [1, 2, 1]], np.int8)
ops = {"copy": lambda d, s: g.probe_strip_copy(d, s), "sobel": lambda d, s: g.sobel_batch(d, s), "sobel-nokeep": lambda d, s: g.sobel_batch(d, s),
       "blur1": lambda d, s: g.blur_batch(d, s, 1), "blur2": lambda d, s: g.blur_batch(d, s, 2), "erode": lambda d, s: g.erode_batch(d, s), "dilate": lambda d, s: g.dilate_batch(d, s),
       "blur2+sobel": lambda d, s: g.blur_sobel_batch(d, s, 2)}
print("tag", os.environ.get("UB_TAG", "default"))
print("%-12s" % "op/off(src,dst)" + "".join("%9s" % ("%d,%d" % o) for o in [(0,0),(1,1),(2,2),(4,4),(8,8),(1,0),(0,1),(4,0),(0,4)]))
for name, fn in ops.items():
    g.tune(23, 1 if name == "copy+halo" else 0); g.tune(22, 1 if name == "sobel-nokeep" else 0)
    row = []
    for so, do in [(0,0),(1,1),(2,2),(4,4),(8,8),(1,0),(0,1),(4,0),(0,4)]:
        s = sb[so:so + F * H * W].view(F, H, W); d = db[do:do + F * H * W].view(F, H, W)
        row.append(timeit(lambda: fn(d, s)))
    print("%-12s" % name + "".join("%9.4f" % t for t in row), flush=True)
g.tune(23, 0); g.tune(22, 0)
