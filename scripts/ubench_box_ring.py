#!/usr/bin/env python3
"""k_box16r (window rows in a register ring, radius a template constant, r <= 16) vs k_box16 (any radius; gsh_tune key 6 = 4) on
64 x 3840x2160 and 8 x 1920x1080: gs_blur r = 4..8, gs_adaptive_threshold r = 1..8.  GB/s = 2 B/px / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
SHAPES = ((64, 2160, 3840), (8, 1080, 1920), (32, 720, 1280)) if not os.environ.get("UB_BIG_ONLY") else ((64, 2160, 3840),)
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (F, H, W) in SHAPES:
    src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
    dst = torch.zeros_like(src); ref = torch.zeros_like(src)
    npx = F * W * H
    for name, fn, radii in (("gs_blur", lambda d, r: g.blur_batch(d, src, r), tuple(range(4, 17))),
                            ("gs_adaptive_threshold", lambda d, r: g.adaptive_threshold_batch(d, src, r, 5), tuple(range(1, 17)))):
        for r in radii:
            out = []
            for key in (0, 4):
                g.tune(6, key)
                ms = timeit(lambda: fn(dst if key != 4 else ref, r))
                out.append(ms)
            g.tune(6, 0)
            same = bool(torch.equal(dst, ref))
            print("%d x %dx%d %-22s r=%d  ring %.4f ms (%.2f of 8 TB/s)   any-radius %.4f ms (%.2f)   same bytes: %s"
                  % (F, W, H, name, r, out[0], 2 * npx / out[0] / 1e6 / 8000, out[1], 2 * npx / out[1] / 1e6 / 8000, same), flush=True)
