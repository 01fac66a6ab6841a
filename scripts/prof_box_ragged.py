#!/usr/bin/env python3
"""the sliding-box launches on a ragged and an aligned 4K batch, for rocprofv3 --kernel-trace --stats: how the ragged frame's
time splits between the ring kernel's body launch and k_box_edge"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
for w in (3840, 3838):
    src = torch.empty((64, 2160, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 5)
    dst = torch.zeros_like(src)
    for _ in range(5):
        g.blur_batch(dst, src, 5); g.blur_batch(dst, src, 16); g.adaptive_threshold_batch(dst, src, 15, 5)
    torch.cuda.synchronize()
