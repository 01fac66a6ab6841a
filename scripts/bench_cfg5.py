#!/usr/bin/env python3
"""config 5 per GPU: per frame gs_blur(r=2) -> gs_sobel -> gs_integral(sobel) -> gs_lbp_detect
(sf=1.1, scales 1..4, step 1, max_rects 4096) on 3840x2160 frames; frames/s on ONE GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.lib(); g.use_torch_stream(); g.tune(4, int(os.environ.get("C5_PRESET", "0")))
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
n, h, w = int(os.environ.get("C5_FRAMES", 16)), 2160, 3840
src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
a, b = torch.zeros_like(src), torch.zeros_like(src)
ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda")
rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
dc = g.cascade_create(casc)
def step():
    g.blur_batch(a, src, 2); b.zero_(); g.sobel_batch(b, a); g.integral_batch(b, ii)
    g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1)
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); reps = 3
for _ in range(reps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
nwin = g.lbp_window_count(casc, w, h, 1.1, 1.0, 4.0, 1)
r0 = rects[0, :int(counts[0])].cpu().numpy()
by_scale = {int(k): int(v) for k, v in zip(*np.unique(r0[:, 2], return_counts=True))}
out = {"hits_by_window_size_frame0": by_scale, "frames": n, "ms_per_frame": round(dt / n * 1e3, 3), "frames_per_s_per_gpu": round(n / dt, 1),
       "Mpix/s": round(n * w * h / dt / 1e6, 1), "Gwin/s": round(n * nwin / dt / 1e9, 2), "detections": counts.cpu().tolist()[:4]}
# verify one frame against the oracle chain (slow on CPU: ~25 s/frame/core for the cascade) only when asked
if os.environ.get("C5_VERIFY"):
    from oracle.pyoracle import Oracle
    o = Oracle("port"); img = Oracle.synth(w, h, 1000)
    s = o.sobel(o.blur(img, 2)); r = o.lbp_detect(casc, o.integral(s), 4096, 1.1, 1.0, 4.0, 1)
    got = rects[0, :int(counts[0])].cpu().numpy().view(np.uint32)
    out["parity_frame0"] = bool(int(counts[0]) == len(r) and np.array_equal(got, np.stack([r["x"], r["y"], r["w"], r["h"]], 1)))
print(json.dumps(out))
