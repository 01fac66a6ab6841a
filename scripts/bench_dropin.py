#!/usr/bin/env python3
"""drop-in (host pointer) path: one 3840x2160 frame per call, PCIe-inclusive, vs the reference on one core"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from oracle.pyoracle import Oracle, have_reference
g = gs.lib()
w, h = 3840, 2160
img = Oracle.synth(w, h, 1); a = np.zeros_like(img); b = np.zeros_like(img)
def t(fn, reps=10):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
out = {"frame": "%dx%d" % (w, h)}
out["gs_blur_r2_host_ms"] = round(t(lambda: g.blur(a, img, 2)), 3)
out["gs_sobel_host_ms"] = round(t(lambda: g.sobel(b, a)), 3)
out["gs_otsu_host_ms"] = round(t(lambda: g.otsu_threshold(b)), 3)
out["gs_threshold_host_ms"] = round(t(lambda: g.threshold(b.copy(), 54)), 3)
def chain():
    g.blur(a, img, 2); g.sobel(b, a); g.threshold(b, g.otsu_threshold(b))
out["chain_host_ms"] = round(t(chain), 3)
out["chain_host_Mpix/s"] = round(w * h / out["chain_host_ms"] / 1e3, 1)
# same chain with the image already on the device (zero-copy drop-in calls, one frame per call)
d = torch.from_numpy(img).cuda(); da, db = torch.zeros_like(d), torch.zeros_like(d)
def chain_dev():
    g.blur(da, d, 2); g.sobel(db, da); g.threshold(db, g.otsu_threshold(db))
out["chain_device_ptr_ms"] = round(t(chain_dev, 50), 4)
out["chain_device_ptr_Mpix/s"] = round(w * h / out["chain_device_ptr_ms"] / 1e3, 1)
o = Oracle("reference" if have_reference() else "port")
def chain_ref():
    s = o.sobel(o.blur(img, 2)); o.threshold(s, o.otsu_threshold(s))
out["reference_one_core_ms"] = round(t(chain_ref, 2), 1)
print(json.dumps(out))
