#!/usr/bin/env python3
"""Fixed cost of k_lbp_cascade per frame: one-stage cascades with 1, 2, 3 weak classifiers (time = fixed + m x weak),
prefilter off, 8 x 4K block-noise frames; and the same through the prefilter kernel alone (a 2-stage cascade whose
second stage is trivial)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade, _FIELDS
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
full = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
def one_stage(m, thr=-1e9):
    arrays = {name: getattr(full, name).copy() for name, _ in _FIELDS}
    arrays["stage_weak_start"] = arrays["stage_weak_start"][:1]
    arrays["stage_nweaks"] = np.array([m], np.uint16)
    arrays["stage_threshold"] = np.array([thr], np.float32)
    return Cascade(full.window_w, full.window_h, **arrays)
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
w, h, n = 3840, 2160, 8
src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(src, ii)
rects1 = torch.zeros((1, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
g.tune(14, 0); g.tune(16, 1)
for thr, what in ((1e9, "every window dies in stage 0"), (-1e9, "every window passes")):
    for m in (1, 2, 3):
        dc = g.cascade_create(one_stage(m, thr))
        ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects1, counts, 0x7fffffff, 1.1, 1.0, 4.0, 1)) / n
        print("one stage, %d weak classifier(s), %s: %.3f ms/frame" % (m, what, ms), flush=True)
        dc.close()
g.tune(14, 0); g.tune(16, 0)
