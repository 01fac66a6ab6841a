#!/usr/bin/env python3
"""fixed workload for rocprofv3 --pmc passes over the sliding-box and integral kernels: 64 x 4K, gs_blur r = 8
(k_box16<0>), gs_adaptive_threshold r = 8 (k_box16<1>), gs_integral (colsum / colbase / wave)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
if os.environ.get("BOX_ANY_RADIUS"): g.tune(6, 4)  # the any-radius kernel k_box16 instead of the ring kernels
F, H, W = 64, 2160, 3840
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
ii = torch.zeros((F, H, W), dtype=torch.int32, device="cuda")
for _ in range(3):
    g.blur_batch(dst, src, 8); g.adaptive_threshold_batch(dst, src, 8, 5); g.integral_batch(src, ii)
torch.cuda.synchronize()
