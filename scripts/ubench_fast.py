#!/usr/bin/env python3
"""gsh_fast_batch on 32 x 1280x720: the whole call and its score pass alone (gsh_fast_score_batch) on the inputs the reviews
ask for -- flat, block noise (configs[3]'s synth frame), tiled lena, random bytes -- with the default score kernel
k_fast_score_q4 and, for reference, the generic k_fast_score_px (gsh_tune key 7 = 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from oracle.pyoracle import Oracle
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
W, H, F = int(os.environ.get("UB_W", 1280)), int(os.environ.get("UB_H", 720)), int(os.environ.get("UB_F", 32))
def timeit(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
A = Oracle.synth(W, H, 4)
from tests.util import lena
L = lena()
inputs = {"flat": np.full_like(A, 128), "block noise (synth seed 4)": A,
          "block noise + 24 (no p < t)": np.minimum(A.astype(np.int32) + 24, 255).astype(np.uint8),
          "lena tiled": np.tile(L, ((H + 127) // 128, (W + 127) // 128))[:H, :W].copy(),
          "random bytes": np.random.RandomState(1).randint(0, 256, A.shape).astype(np.uint8)}
kps = torch.zeros((F, 2000, 12), dtype=torch.int32, device="cuda"); cnt = torch.zeros(F, dtype=torch.int32, device="cuda")
for rnd in range(2):
    for name, img in inputs.items():
        src = torch.from_numpy(np.stack([img] * F)).cuda(); sm = torch.zeros_like(src)
        for px in ((0, 2) if rnd == 0 else (0,)):
            g.tune(7, px)
            ms = timeit(lambda: g.fast_batch(src, sm, kps, cnt, 2000, 20))
            ms_score = timeit(lambda: g.fast_score_batch(sm, src, 20))
            print("%-30s %-16s gs_fast %6.1f us per %d frames (%.2f us per frame, %.0f Gpx/s)  score pass alone %6.1f us  n0=%d"
                  % (name, ("k_fast_score_q4", "", "k_fast_score_px")[px], ms * 1e3, F, ms * 1e3 / F, F * W * H / ms / 1e6, ms_score * 1e3, int(cnt[0])), flush=True)
        g.tune(7, 0)
