#!/usr/bin/env python3
"""small fixed workload for rocprofv3 --pmc passes: each strip kernel a few times on 64 4K frames"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()  # UB_LIB=build_variants/libgs_experiment.so for the strip-copy probe
F, H, W = 64, 2160, 3840
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src); tmp = torch.zeros_like(src)
hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda"); thr = torch.zeros(F, dtype=torch.uint8, device="cuda")
for _ in range(3):
    g.blur_batch(tmp, src, 2); g.sobel_batch(dst, tmp); g.erode_batch(dst, src)
    g.otsu_batch(dst, hist, thr); g.threshold_batch(dst, thr)
    g.edge_pipeline_batch(dst, None, src, 2, hist, thr)
# the "next" rows with their own kernels + the barrier-free integral
import numpy as np
ii = torch.zeros((F, H, W), dtype=torch.int32, device="cuda")
half = torch.zeros((F, H // 2, W // 2), dtype=torch.uint8, device="cuda")
gauss = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.int8)
for _ in range(3):
    g.filter_batch(dst, src, gauss, 16); g.adaptive_threshold_batch(dst, src, 15, 5); g.blur_batch(dst, src, 9)
    g.downsample_batch(half, src); g.integral_batch(src, ii)
# gs_fast on 32 x 720p block-noise frames (configs[3]): score pass 1 R + 1 W, NMS 1 R
f7 = torch.empty((32, 720, 1280), dtype=torch.uint8, device="cuda"); g.synth_batch(f7, 4)
sm7 = torch.zeros_like(f7)
kp7 = torch.zeros((32, 2000, 12), dtype=torch.int32, device="cuda"); cn7 = torch.zeros(32, dtype=torch.int32, device="cuda")
for _ in range(3):
    g.fast_batch(f7, sm7, kp7, cn7, 2000, 20)
torch.cuda.synchronize()
print("algorithmic bytes per launch: copy/blur/erode/threshold %d, sobel %d, hist %d" % (2*F*H*W, F*(H*W+(H-2)*(W-2)), F*H*W))
