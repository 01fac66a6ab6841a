#!/usr/bin/env python3
"""gs_match_template on device buffers: matrix-core kernel (k_match_template_mfma, default) vs the dot-product kernels
(gsh_tune key 20 = 1); taps per second = rw * rh * tw * th / time; both outputs compared, one slab against the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from oracle.pyoracle import Oracle
g = gs.lib(); g.use_torch_stream()
o = Oracle("port")
def timeit(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (iw, ih) in ((1280, 720), (3840, 2160)):
    img = Oracle.synth(iw, ih, 4)
    d_img = torch.from_numpy(img).cuda()
    for (tw, th) in ((16, 16), (32, 32), (64, 64), (128, 128), (181, 181), (256, 64)):
        t = img[100:100 + th, 200:200 + tw].copy(); t[::3, ::5] ^= 0x55
        d_t = torch.from_numpy(t).cuda()
        rw, rh = iw - tw + 1, ih - th + 1
        out = []
        res = []
        for key in (0, 1):
            g.tune(20, key)
            r = torch.zeros((rh, rw), dtype=torch.uint8, device="cuda")
            ms = timeit(lambda: g.match_template(d_img, d_t, r))
            out.append(ms); res.append(r)
        g.tune(20, 0)
        same = bool(torch.equal(res[0], res[1]))
        slab = o.match_template(img[90:90 + th + 7, :], t) if iw <= 1280 else None
        ok = None if slab is None else bool(np.array_equal(res[0][90:98].cpu().numpy(), slab))
        taps = rw * rh * tw * th
        print("%dx%d template %dx%d: mfma %.4f ms (%.1f Ttap/s)   dot4 %.4f ms (%.1f Ttap/s)   same bytes: %s   8 rows == oracle: %s"
              % (iw, ih, tw, th, out[0], taps / out[0] / 1e9, out[1], taps / out[1] / 1e9, same, ok), flush=True)
