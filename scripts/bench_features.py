#!/usr/bin/env python3
"""secondary measurements: configs 3-5 of BASELINE.json (integral + LBP cascade, FAST/ORB/match)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
def timeit(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
# ---- config 3: integral + lbp_detect on 1080p (and 4K), sf=1.1, scales 1..4, step 1
for (w, h, n, seed) in ((1920, 1080, 8, 3), (3840, 2160, 4, 1000)):
  for variant in (0, 1):
      g.tune(6, variant)
      src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, seed)
      ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda")
      ms_int = timeit(lambda: g.integral_batch(src, ii))
      dc = g.cascade_create(casc)
      rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
      nwin = g.lbp_window_count(casc, w, h, 1.1, 1.0, 4.0, 1)
      ms_lbp = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1), 3)
      out["cfg3_%dx%d_%s" % (w, h, "generic" if variant else "banded")] = {"frames": n, "integral_ms_per_frame": round(ms_int / n, 4), "integral_GBs_5Bpx": round(5.0 * n * w * h / ms_int / 1e6, 1),
                                   "lbp_ms_per_frame": round(ms_lbp / n, 3), "windows_per_frame": nwin, "Mwin/s": round(nwin * n / ms_lbp / 1e3, 1),
                                   "detections": counts.cpu().tolist()}
      dc.close()
g.tune(6, 0)
# ---- config 4: ORB extract + match on a 1280x720 pair
from oracle.pyoracle import Oracle
A = Oracle.synth(1280, 720, 4); B = np.zeros_like(A); B[:717, :1275] = A[3:, 5:]
dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(); sm = torch.zeros_like(dA)
t0 = time.perf_counter(); reps = 10
for _ in range(reps): ka = g.orb_extract_dev(dA, sm, 500, 20)
t_orb = (time.perf_counter() - t0) / reps
kb = g.orb_extract_dev(dB, sm, 500, 20)
t0 = time.perf_counter()
for _ in range(reps): m = g.match_orb(ka, kb, 2500, 60.0)
t_match = (time.perf_counter() - t0) / reps
src = torch.from_numpy(np.stack([A] * 32)).cuda(); smb = torch.zeros_like(src)
kps = torch.zeros((32, 5000, 12), dtype=torch.int32, device="cuda"); cnt = torch.zeros(32, dtype=torch.int32, device="cuda")
ms_fast = timeit(lambda: g.fast_batch(src, smb, kps, cnt, 5000, 20))
out["cfg4_1280x720"] = {"orb_extract_ms(host-sync incl.)": round(t_orb * 1e3, 3), "keypoints": len(ka), "match_500x500_ms(host path)": round(t_match * 1e3, 3), "matches": len(m),
                        "fast_batch32_ms_per_frame": round(ms_fast / 32, 4), "fast_GBs_3Bpx": round(3.0 * 32 * 1280 * 720 / ms_fast / 1e6, 1)}
print(json.dumps(out))
