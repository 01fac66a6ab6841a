import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
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
for nt in (256, 128):
    for rows in (32, 48, 64):
        g.tune(26, nt); g.tune(25, rows)
        print("threads %d rows %d: score noise %.1f flat %.1f | gs_fast noise %.1f flat %.1f" % (nt, rows, timeit(lambda: g.probe_fast_score(sm, f, 20)), timeit(lambda: g.probe_fast_score(sm, flat, 20)),
              timeit(lambda: g.fast_batch(f, sm, kp, cn, 2000, 20)), timeit(lambda: g.fast_batch(flat, sm, kp, cn, 2000, 20))), flush=True)
