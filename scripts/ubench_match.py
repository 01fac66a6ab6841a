#!/usr/bin/env python3
"""gs_match_orb device-resident (gsh_match_orb_dev) at 500 x 500, 2500 x 2500 and 10000 x 10000 descriptors (random bits), wall
time per call by stream events"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
torch.manual_seed(1)
for nk in (500, 2500, 10000):
    kk = torch.randint(-2**31, 2**31 - 1, (2, nk, 12), dtype=torch.int64, device="cuda").to(torch.int32)
    kk[1, : nk // 2, 4:] = kk[0, : nk // 2, 4:]  # half of the queries have an exact partner
    mt = torch.zeros((nk, 3), dtype=torch.int32, device="cuda"); mc = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rnd in range(2):
        ms = timeit(lambda: g.match_orb_dev(kk[0], nk, kk[1], nk, mt, mc, nk, 60.0))
        print("%5d x %5d: %.4f ms  %.1f Gpairs/s  matches %d" % (nk, nk, ms, nk * nk / ms / 1e6, int(mc[0])), flush=True)
