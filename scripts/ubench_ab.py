#!/usr/bin/env python3
"""within-process A/B of library variants (build_variants/libgs_<tag>.so) on the strip kernels"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = os.environ.get("AB_TAGS", "base").split(",")
libs = {t: (gs.lib() if t == "base" else gs.Grayskull(os.path.join(ROOT, "build_variants", "libgs_%s.so" % t))) for t in tags}
for g in libs.values(): g.use_torch_stream()
W, H, F = int(os.environ.get("UB_W", 3840)), int(os.environ.get("UB_H", 2160)), int(os.environ.get("UB_F", 64))
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); libs[tags[0]].synth_batch(src, 1000)
dst = torch.zeros_like(src); npx = F * W * H
def timeit(fn, reps=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
Ts = [int(x) for x in os.environ.get("UB_T", "0").split(",")]
ops = os.environ.get("UB_OPS", "copy,erode,sobel,blur2").split(",")
def call(g, op):
    return {"copy": lambda: g.probe_strip_copy(dst, src), "erode": lambda: g.erode_batch(dst, src),
            "sobel": lambda: g.sobel_batch(dst, src), "blur2": lambda: g.blur_batch(dst, src, 2),
            "thr": lambda: g.threshold_batch(dst, 100), "dilate": lambda: g.dilate_batch(dst, src),
            "blur1": lambda: g.blur_batch(dst, src, 1), "blur3": lambda: g.blur_batch(dst, src, 3),
            "fused": lambda: g.edge_pipeline_batch(dst, None, src, 2, hist, thr), "bs": lambda: g.blur_sobel_batch(dst, src, 2), "hist": lambda: g.histogram_batch(src, hist)}[op]
hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda")
thr = torch.zeros((F,), dtype=torch.uint8, device="cuda")
res = {}
for rnd in range(int(os.environ.get("AB_ROUNDS", 5))):
    for op in ops:
        for T in Ts:
            for t in tags:
                g = libs[t]; g.tune(0, T); g.tune(1, int(os.environ.get("UB_G", 1)))
                res.setdefault((op, T, t), []).append(timeit(call(g, op)))
print("%-6s %4s %-10s %9s %9s %8s" % ("op", "T", "variant", "ms(med)", "ms(min)", "GB/s"))
for (op, T, t), v in sorted(res.items()):
    ms = float(np.median(v)); b = (1.0 if op == "hist" else 2.0) * npx
    print("%-6s %4d %-10s %9.4f %9.4f %8.1f" % (op, T, t, ms, min(v), b / ms / 1e6))
