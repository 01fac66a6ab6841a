#!/usr/bin/env python3
"""fixed workload for rocprofv3 --pmc passes over the fused kernels only: 32-frame launches (the
pipeline's chunk size) of k_blur_sobel_hist16<2, true> and <2, false>, plus the strip copy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()  # UB_LIB=build_variants/libgs_experiment.so for the strip-copy probe
F, H, W = 32, 2160, 3840
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda"); thr = torch.zeros(F, dtype=torch.uint8, device="cuda")
for _ in range(4):
    if g.experiment: g.probe_strip_copy(dst, src)  # experiment builds only
    g.blur_sobel_batch(dst, src, 2)
    g.edge_pipeline_batch(dst, None, src, 2, hist, thr)
torch.cuda.synchronize()
