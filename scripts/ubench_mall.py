#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache serve reads of data a kernel has just WRITTEN?  gs_threshold (in place: 1 R + 1 W per px) on n 4K
frames right after gs_sobel wrote them ("warm") vs after 600 MB of other traffic ("cold"); run under
`rocprofv3 --kernel-trace` and read the k_threshold durations (scripts/gpu_mall.sh) -- event timing of single launches is
dominated by the ~35 us of launch + event latency.  UB_TAG picks a library variant (st0 = cache-allocating stores).
If warm were much faster, a threshold pass running a few frames behind the fused blur+sobel kernel could skip its HBM reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("UB_TAG", "")
g = gs.Grayskull(os.path.join(ROOT, "build_variants", "libgs_%s.so" % TAG)) if TAG else gs.lib()
g.use_torch_stream()
W, H = 3840, 2160
big = torch.empty((72, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(big, 7)
big2 = torch.empty_like(big)
for n in (2, 4, 8, 16, 32):
    src = big[:n]; dst = torch.empty_like(src)
    for rep in range(4):
        g.sobel_batch(dst, src); g.threshold_batch(dst, 100 + n)           # warm: threshold value tags the launch (n)
        g.sobel_batch(dst, src); g.blur_batch(big2, big, 2); g.threshold_batch(dst, 200 + n)  # cold
    torch.cuda.synchronize()
