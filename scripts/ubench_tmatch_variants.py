#!/usr/bin/env python3
"""timing experiment: k_match_template_mfma without its MFMAs / without its in-loop operand loads (results are wrong on purpose)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from oracle.pyoracle import Oracle
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
def timeit(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (iw, ih) in ((1280, 720), (3840, 2160)):
    img = Oracle.synth(iw, ih, 4); d_img = torch.from_numpy(img).cuda()
    for (tw, th) in ((64, 64), (128, 128)):
        t = torch.from_numpy(img[100:100 + th, 200:200 + tw].copy()).cuda()
        r = torch.zeros((ih - th + 1, iw - tw + 1), dtype=torch.uint8, device="cuda")
        print("%s %dx%d template %dx%d: %.4f ms" % (os.environ.get("UB_LIB", "default")[-14:], iw, ih, tw, th, timeit(lambda: g.match_template(d_img, t, r))), flush=True)
