#!/usr/bin/env python3
"""gsh_fast_batch: pass 2 as the strip kernel k_fast_nms16 (default) vs the item-by-item kernel k_fast_nms (gsh_tune key 19 = 1),
32 x 1280x720, threshold 20; identical keypoints checked.  Also 8 x 4K and 64 x 480p."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (W, H, F, cap) in ((1280, 720, 32, 2000), (1280, 720, 32, 5000), (3840, 2160, 8, 5000), (640, 480, 64, 2000)):
    src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 4)
    flat = torch.full_like(src, 128)
    kps = torch.zeros((F, cap, 12), dtype=torch.int32, device="cuda"); cnt = torch.zeros(F, dtype=torch.int32, device="cuda")
    for name, img in (("synth", src), ("flat", flat)):
        sm = torch.zeros_like(img)
        ref = None
        for rnd in range(2):
            for k19 in (1, 0):
                g.tune(19, k19)
                ms = timeit(lambda: g.fast_batch(img, sm, kps, cnt, cap, 20))
                crc = zlib.crc32(kps.cpu().numpy().tobytes()) ^ zlib.crc32(cnt.cpu().numpy().tobytes())
                ref = ref or crc
                print("%dx%d x%d cap %d %-6s NMS %-14s %.4f ms per frame (%.1f us per batch)  n0=%d same=%s"
                      % (W, H, F, cap, name, "item kernel" if k19 else "sparse (default)", ms / F, ms * 1e3, int(cnt[0]), crc == ref), flush=True)
        g.tune(19, 0)
