#!/bin/bash
# round 4, visit l: gs_fast -- tile height of the score kernel (key 25 = 16 / 32 / 48 / 64) and the NMS pass that skips empty
# row spans; ragged sliding box after the cheaper tail path
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04l_fast_tile_rows.log
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, grayskull_amd as gs
from tests.util import lena
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nf, h, w = 32, 720, 1280
frames = {}
f = torch.empty((nf, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(f, 4); frames["block noise (configs[3])"] = f
frames["flat"] = torch.full((nf, h, w), 100, dtype=torch.uint8, device="cuda")
l = np.tile(lena(), (h // 128 + 1, w // 128 + 1))[:h, :w]; frames["tiled lena"] = torch.from_numpy(np.stack([l] * nf)).cuda()
frames["random bytes"] = torch.randint(0, 256, (nf, h, w), dtype=torch.uint8, device="cuda")
frames["bright noise"] = (torch.randint(0, 16, (nf, h, w), dtype=torch.uint8, device="cuda") + 200)
sm = torch.zeros((nf, h, w), dtype=torch.uint8, device="cuda")
kp = torch.zeros((nf, 2000, 12), dtype=torch.int32, device="cuda"); cn = torch.zeros(nf, dtype=torch.int32, device="cuda")
print("%-26s %6s %12s %12s   counts[0]" % ("frames (32 x 720p)", "rows", "score us", "gs_fast us"))
for name, fr in frames.items():
    ref = None
    for rows in (16, 32, 48, 64):
        g.tune(25, rows)
        ts = timeit(lambda: g.probe_fast_score(sm, fr, 20)); tf = timeit(lambda: g.fast_batch(fr, sm, kp, cn, 2000, 20))
        torch.cuda.synchronize(); sig = (int(cn.sum()), int(kp.sum()))
        ref = ref or sig
        print("%-26s %6d %12.1f %12.1f   %d %s" % (name, rows, ts, tf, int(cn[0]), "" if sig == ref else "MISMATCH"), flush=True)
g.tune(25, 0)
# 4K and 1080p frames
for (hh, ww, n) in ((2160, 3840, 8), (1080, 1920, 8)):
    fr = torch.empty((n, hh, ww), dtype=torch.uint8, device="cuda"); g.synth_batch(fr, 4)
    s2 = torch.zeros_like(fr); k2 = torch.zeros((n, 5000, 12), dtype=torch.int32, device="cuda"); c2 = torch.zeros(n, dtype=torch.int32, device="cuda")
    for rows in (16, 32, 64):
        g.tune(25, rows); print("%dx%d x%d rows %d: gs_fast %.1f us" % (ww, hh, n, rows, timeit(lambda: g.fast_batch(fr, s2, k2, c2, 5000, 20))), flush=True)
g.tune(25, 0)
PY
echo "== ragged box"; RG_CHECK=0 timeout 500 python scripts/ubench_ragged.py 2>&1 | grep -v amdgpu.ids | grep -E "^(blur r|adaptive|op )" | tee gpurun_out/r04l_ragged_box.log
