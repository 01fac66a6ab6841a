#!/usr/bin/env python3
"""micro-benchmark sweep of the strip kernels' launch tuning (run on the GPU box).
Interleaved rounds in ONE process; prints median GB/s (algorithmic 2 B/px) per variant."""
import json, os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs

g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib()
g.use_torch_stream()
W, H, F = int(os.environ.get("UB_W", 3840)), int(os.environ.get("UB_H", 2160)), int(os.environ.get("UB_F", 64))
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
npx = F * W * H

def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

ops = {"copy": lambda: g.probe_strip_copy(dst, src), "erode": lambda: g.erode_batch(dst, src),
       "sobel": lambda: g.sobel_batch(dst, src), "blur2": lambda: g.blur_batch(dst, src, 2),
       "blur1": lambda: g.blur_batch(dst, src, 1), "blur3": lambda: g.blur_batch(dst, src, 3)}
which = os.environ.get("UB_OPS", "copy,erode,sobel,blur2").split(",")
Ts = [int(x) for x in os.environ.get("UB_T", "0,16,32,64,128,270").split(",")]
geoms = [int(x) for x in os.environ.get("UB_G", "0,1,2").split(",")]
pfs = [int(x) for x in os.environ.get("UB_PF", "1,2,3").split(",")]
res = {}
for rnd in range(3):
    for op in which:
        for T, gm, pf in itertools.product(Ts, geoms, pfs):
            g.tune(0, T); g.tune(1, gm); g.tune(2, pf)
            res.setdefault((op, T, gm, pf), []).append(timeit(ops[op], 5))
print("%-6s %4s %2s %2s %9s %8s %6s" % ("op", "T", "g", "pf", "ms(med)", "GB/s", "frac"))
best = {}
for (op, T, gm, pf), v in sorted(res.items()):
    ms = float(np.median(v)); gbs = 2.0 * npx / ms / 1e6
    print("%-6s %4d %2d %2d %9.4f %8.1f %6.3f" % (op, T, gm, pf, ms, gbs, gbs / 8000))
    if op not in best or ms < best[op][0]: best[op] = (ms, T, gm, pf, gbs)
print("BEST", json.dumps({k: {"ms": round(v[0], 4), "T": v[1], "geom": v[2], "pf": v[3], "GB/s": round(v[4], 1)} for k, v in best.items()}))
# reference points: plain streaming kernels
t = timeit(lambda: g.threshold_batch(dst, 100)); print("threshold(in place) GB/s %.1f" % (2.0 * npx / t / 1e6))
t = timeit(lambda: dst.copy_(src)); print("torch copy_ GB/s %.1f" % (2.0 * npx / t / 1e6))
t = timeit(lambda: dst.zero_()); print("torch zero_ GB/s %.1f (write only)" % (1.0 * npx / t / 1e6))
