#!/usr/bin/env python3
"""fixed workload for rocprofv3 --pmc passes over the LBP kernels: gs_lbp_detect on 8 x 4K block-noise frames (LBP_EDGE=1: the
edge maps of configs[4]), LBP_MODE = gsh_tune key 14 (0 the rule, 1 k_lbp_cascade for every scale, 2 + i tile shape i) -- two calls"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
n, h, w = int(os.environ.get("LBP_N", 8)), 2160, 3840
fr = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(fr, 1000)
if os.environ.get("LBP_EDGE") == "1":  # the configs[4] input: gs_blur(2) -> gs_sobel into a zeroed image
    a = torch.empty_like(fr); g.blur_batch(a, fr, 2); fr.zero_(); g.sobel_batch(fr, a); del a
ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(fr, ii)
dc = g.cascade_create(Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin")))
rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
g.tune(14, int(os.environ.get("LBP_MODE", 0)))
if os.environ.get("LBP_ONE_LANE") == "1": g.tune(17, 1)  # one lane per re-packed window (round-2 survivors)
for _ in range(2):
    g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)
torch.cuda.synchronize()
dc.close()
