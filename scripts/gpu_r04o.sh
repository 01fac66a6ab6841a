#!/bin/bash
# round 4, visit o (second run: + the sparse NMS pass behind the score kernel's bitmap):
# kernel's queue; per-kernel breakdown of a gs_fast call (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04o_fast.log
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nf, h, w = 32, 720, 1280
f = torch.empty((nf, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(f, 4)
flat = torch.full((nf, h, w), 100, dtype=torch.uint8, device="cuda")
rnd = torch.randint(0, 256, (nf, h, w), dtype=torch.uint8, device="cuda")
sm = torch.zeros((nf, h, w), dtype=torch.uint8, device="cuda")
kp = torch.zeros((nf, 2000, 12), dtype=torch.int32, device="cuda"); cn = torch.zeros(nf, dtype=torch.int32, device="cuda")
for key19 in (0, 2, 0, 2):
    g.tune(19, key19)
    print("NMS %s: " % ("sparse" if key19 == 0 else "strips"), end="")
    print("score noise %.1f flat %.1f random %.1f | gs_fast noise %.1f flat %.1f random %.1f" % (timeit(lambda: g.probe_fast_score(sm, f, 20)), timeit(lambda: g.probe_fast_score(sm, flat, 20)), timeit(lambda: g.probe_fast_score(sm, rnd, 20)),
          timeit(lambda: g.fast_batch(f, sm, kp, cn, 2000, 20)), timeit(lambda: g.fast_batch(flat, sm, kp, cn, 2000, 20)), timeit(lambda: g.fast_batch(rnd, sm, kp, cn, 2000, 20))), flush=True)
PY
cat > /tmp/fast_loop.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
f = torch.empty((32, 720, 1280), dtype=torch.uint8, device="cuda"); g.synth_batch(f, 4)
sm = torch.zeros_like(f); kp = torch.zeros((32, 2000, 12), dtype=torch.int32, device="cuda"); cn = torch.zeros(32, dtype=torch.int32, device="cuda")
for _ in range(50): g.fast_batch(f, sm, kp, cn, 2000, 20)
torch.cuda.synchronize()
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04o_prof -o fast -- python /tmp/fast_loop.py > $R/gpurun_out/r04o_prof.log 2>&1
cd $R; f=$(find gpurun_out/r04o_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04o_fast_kernel_stats.csv && head -12 gpurun_out/r04o_fast_kernel_stats.csv | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py tests/test_ragged.py -q -m gpu -k "fast or orb or kat or match or pyramid" 2>&1 | tail -3 | tee gpurun_out/r04o_pytest.log
