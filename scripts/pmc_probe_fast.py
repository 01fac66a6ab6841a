#!/usr/bin/env python3
"""gs_fast on 32 x 720p block-noise frames for --pmc passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
f7 = torch.empty((32, 720, 1280), dtype=torch.uint8, device="cuda"); g.synth_batch(f7, 4)
sm7 = torch.zeros_like(f7)
kp7 = torch.zeros((32, 2000, 12), dtype=torch.int32, device="cuda"); cn7 = torch.zeros(32, dtype=torch.int32, device="cuda")
for k in (6, 0):  # both passes in one walk / two passes
    g.tune(7, k)
    for _ in range(3):
        g.fast_batch(f7, sm7, kp7, cn7, 2000, 20)
torch.cuda.synchronize()
