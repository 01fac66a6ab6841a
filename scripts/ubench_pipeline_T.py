#!/usr/bin/env python3
"""config-2 pipeline (512 4K frames) vs the band height of the fused kernel (gsh_tune key 0; 0 = auto: 24 bands of 90 rows per
32-frame chunk, every wave resident at once)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
F, H, W = 512, 2160, 3840
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda"); thr = torch.zeros(F, dtype=torch.uint8, device="cuda")
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for rnd in range(2):
    for T in [int(x) for x in os.environ.get("UB_T", "0,24,30,36,45,60,72").split(",")]:
        g.tune(0, T)
        ms = timeit(lambda: g.edge_pipeline_batch(dst, None, src, 2, hist, thr))
        cs = (int(dst.view(torch.int32).sum().item()) & 0xffffffff, int(thr.sum()))
        ref = ref or cs
        print("fused band height %3d: %.4f ms  %.0f Mpix/s %s" % (T, ms, F * W * H / ms / 1e3, "ok" if cs == ref else "MISMATCH"))
g.tune(0, 0)
