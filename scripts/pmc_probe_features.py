#!/usr/bin/env python3
"""fixed workload for rocprofv3 --pmc passes over the feature kernels: gs_fast on 32 x 720p block-noise frames (configs[3]),
gs_histogram on 64 x 4K, gs_lbp_detect on 4 x 1080p (configs[2])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.lib(); g.use_torch_stream()
src = torch.empty((32, 720, 1280), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 4)
sm = torch.zeros_like(src)
kps = torch.zeros((32, 5000, 12), dtype=torch.int32, device="cuda"); cnt = torch.zeros(32, dtype=torch.int32, device="cuda")
big = torch.empty((64, 2160, 3840), dtype=torch.uint8, device="cuda"); g.synth_batch(big, 1000)
hist = torch.zeros((64, 256), dtype=torch.int32, device="cuda")
fr = torch.empty((4, 1080, 1920), dtype=torch.uint8, device="cuda"); g.synth_batch(fr, 1000)
ii = torch.zeros((4, 1080, 1920), dtype=torch.int32, device="cuda"); g.integral_batch(fr, ii)
dc = g.cascade_create(Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin")))
rects = torch.zeros((4, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(4, dtype=torch.int32, device="cuda")
for _ in range(3):
    g.fast_batch(src, sm, kps, cnt, 5000, 20)
    g.tune(7, 2); g.fast_score_batch(sm, src, 20); g.tune(7, 0)
    g.histogram_batch(big, hist)
    g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)
torch.cuda.synchronize()
dc.close()
