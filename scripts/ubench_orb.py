#!/usr/bin/env python3
"""gs_orb_extract on 32 x 1280x720 block-noise frames (configs[3]): the GS_NO_STDLIB flavour (all on the device) and the libm
flavour (trig on the host), per batch and per frame, next to gs_fast alone -- what the selection / description steps add"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
F, H, W, NK = 32, 720, 1280, 500
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 4)
sm = torch.zeros_like(src)
kps = torch.zeros((F, NK, 12), dtype=torch.int32, device="cuda"); cnt = torch.zeros(F, dtype=torch.int32, device="cuda")
kf = torch.zeros((F, 2000, 12), dtype=torch.int32, device="cuda")
for rnd in range(2):
    t_fast = timeit(lambda: g.fast_batch(src, sm, kf, cnt, 2000, 20))
    t_ns = timeit(lambda: g.orb_extract_batch_nostdlib(src, sm, kps, cnt, NK, 20))
    t_lm = timeit(lambda: g.orb_extract_batch_dev(src, sm, NK, 20), reps=10)
    print("32 x 720p: gs_fast %.1f us | gs_orb_extract GS_NO_STDLIB %.1f us (%.2f per frame) | libm %.1f us (%.2f per frame)  n0=%d"
          % (t_fast, t_ns, t_ns / F, t_lm, t_lm / F, int(cnt[0])), flush=True)
