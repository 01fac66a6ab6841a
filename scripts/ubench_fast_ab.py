#!/usr/bin/env python3
"""gs_fast on 32 x 720p (block noise / flat / random bytes): score pass and whole call, for the library named by UB_LIB
(a build_variants/ path) or the in-tree one -- one library per process, run the variants alternately on the same box"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
def timeit(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nf, h, w = 32, 720, 1280
f = torch.empty((nf, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(f, 4)
flat = torch.full((nf, h, w), 100, dtype=torch.uint8, device="cuda")
sm = torch.zeros((nf, h, w), dtype=torch.uint8, device="cuda")
kp = torch.zeros((nf, 2000, 12), dtype=torch.int32, device="cuda"); cn = torch.zeros(nf, dtype=torch.int32, device="cuda")
for _ in range(2):
    print("%-34s score us: noise %.1f flat %.1f | gs_fast us: noise %.1f flat %.1f" % (os.environ.get("UB_LIB") or "in-tree",
          timeit(lambda: g.probe_fast_score(sm, f, 20)), timeit(lambda: g.probe_fast_score(sm, flat, 20)),
          timeit(lambda: g.fast_batch(f, sm, kp, cn, 2000, 20)), timeit(lambda: g.fast_batch(flat, sm, kp, cn, 2000, 20))), flush=True)
