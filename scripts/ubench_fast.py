#!/usr/bin/env python3
"""gsh_fast_batch: two passes with the LDS-tile score kernel (default) / both passes in one walk (k_fast_fused, key 7 = 6; its "score pass alone" column is k_fast_score_q4) / with the block-local candidate queue (key 7 = 3) vs strip kernel k_fast_score4 (gsh_tune key 7 = 1) vs the per-pixel kernel with one
global byte load per ring pixel (key 7 = 2), 32 x 1280x720"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from oracle.pyoracle import Oracle
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
W, H, F = int(os.environ.get("UB_W", 1280)), int(os.environ.get("UB_H", 720)), int(os.environ.get("UB_F", 32))
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
A = Oracle.synth(W, H, 4)
inputs = {"synth": A, "synth_bright(+24, no p<t)": np.minimum(A.astype(np.int32) + 24, 255).astype(np.uint8),
          "flat": np.full_like(A, 128), "lena_tiled": None, "random": np.random.RandomState(1).randint(0, 256, A.shape).astype(np.uint8)}
kps = torch.zeros((F, 5000, 12), dtype=torch.int32, device="cuda"); cnt = torch.zeros(F, dtype=torch.int32, device="cuda")
from tests.util import lena
L = lena(); inputs["lena_tiled"] = np.tile(L, ((H + 127) // 128, (W + 127) // 128))[:H, :W].copy()
if os.environ.get("UB_SYNTH_ONLY"): inputs = {"synth": inputs["synth"]}
if os.environ.get("UB_M"): g.tune(0, int(os.environ["UB_M"]))
for name, img in inputs.items():
    src = torch.from_numpy(np.stack([img] * F)).cuda(); sm = torch.zeros_like(src)
    for px in ((0, 6, 4, 3, 1, 2) if not os.environ.get("UB_ONLY") else tuple(int(v) for v in os.environ["UB_ONLY"].split(","))):
        g.tune(7, px); g.tune(18, int(os.environ.get("UB_K18", 0)))
        ms = timeit(lambda: g.fast_batch(src, sm, kps, cnt, 5000, 20))
        ms_score = timeit(lambda: g.probe_fast_score(sm, src, 20))
        print("%-28s %-10s %.4f ms per frame  (%.0f Gpx/s)  score pass alone %.1f us per batch  n0=%d"
              % (name, ("tile4+queue", "strip", "px", "tile+queue", "tile", "-", "fused walk")[px], ms / F, F * W * H / ms / 1e6, ms_score * 1e3, int(cnt[0])), flush=True)
    g.tune(7, 0)
