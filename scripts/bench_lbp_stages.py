#!/usr/bin/env python3
"""Where the cascade's time goes: gs_lbp_detect with the frontalface cascade TRUNCATED to its first k stages
(k = 1 .. 20), with the rule's kernels (gsh_tune key 14 = 0: k_lbp_tile where its tile fits) and with k_lbp_cascade for every scale
(key 14 = 1), 8 x 4K block-noise frames and 8 x 4K edge maps.  Needs the experiment library (UB_LIB=build_variants/libgs_experiment.so:
key 16 drops the rect emission so that a cap that is never reached needs no rect buffer).
time(k) - time(k-1) = what stage k-1 costs; `alive` = detections of the truncated cascade / windows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade, _FIELDS
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
full = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
def truncated(k):
    arrays = {name: getattr(full, name) for name, _ in _FIELDS}
    for name in ("stage_weak_start", "stage_nweaks", "stage_threshold"):
        arrays[name] = arrays[name][:k]
    return Cascade(full.window_w, full.window_h, **arrays)
def timeit(fn, reps=2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,5,6,8,10,14,20").split(",")]
pres = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
w, h, n = 3840, 2160, 8
src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
a, b = torch.empty_like(src), torch.zeros_like(src)
g.blur_batch(a, src, 2); g.sobel_batch(b, a)
nwin = g.lbp_window_count(full, w, h, 1.1, 1.0, 4.0, 1)
cap = 0x7fffffff  # never reached: nothing is skipped; key 16 = 1 drops the rect emission so no rect buffer of that size is needed
rects1 = torch.zeros((1, nwin // 8, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
for name, img in (("noise", src), ("edges", b)):
    ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(img, ii)
    for pre in pres:
        g.tune(14, pre)
        prev = 0.0
        for k in ks:
            dc = g.cascade_create(truncated(k))
            g.tune(16, 1)
            ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects1, counts, cap, 1.1, 1.0, 4.0, 1)) / n
            g.tune(16, 0)
            ev = torch.zeros(4, dtype=torch.int64, device="cuda"); g.lbp_count_evaluated(ev)
            g.tune(16, 1); g.lbp_detect_batch(dc, ii[:1], rects1, counts[:1], cap, 1.1, 1.0, 4.0, 1); g.tune(16, 0)
            torch.cuda.synchronize(); g.lbp_count_evaluated(None)
            wk = float(ev[1]) / nwin
            alive = -1.0
            if k >= 3:  # pass fraction of the truncated cascade on frame 0 (untimed; the rect buffer holds nwin / 8)
                g.lbp_detect_batch(dc, ii[:1], rects1, counts[:1], nwin // 8, 1.1, 1.0, 4.0, 1)
                alive = float(counts[0]) / nwin
            print("%s 4K x%d prefilter %2d  stages %2d  %.3f ms/frame (+%.3f)  weak evals/window %.3f  pass fraction %.5f"
                  % (name, n, pre, k, ms, ms - prev, wk, alive), flush=True)
            prev = ms
            dc.close()
    g.tune(14, 0)
