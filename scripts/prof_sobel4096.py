#!/usr/bin/env python3
"""north-star shape in isolation, for rocprofv3 --kernel-trace --stats: gs_sobel alone on 64 distinct 4096 x 4096 frames
(1 GiB per plane), 3 untimed + 20 launches.  Prints the HIP-event average next to what the trace will show."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
n = 64
a = torch.empty((n, 4096, 4096), dtype=torch.uint8, device="cuda"); g.synth_batch(a, 2)
b = torch.zeros_like(a)
for _ in range(3): g.sobel_batch(b, a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.sobel_batch(b, a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
by = float(n * (4096 * 4096 + 4094 * 4094))
print("gs_sobel 64 x 4096x4096: %.4f ms per launch (HIP events), %.1f GB/s algorithmic = %.4f of 8 TB/s, %.0f Mpix/s" % (ms, by / ms / 1e6, by / ms / 1e6 / 8000, n * 4096 * 4096 / ms / 1e3))
