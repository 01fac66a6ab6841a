#!/bin/bash
# round 4, visit h: A/B of the fused blur+sobel kernel held to 128 registers (4 waves per SIMD, 72 B of scratch) against 138
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" build_variants/libgs_minw4.so; do
  echo "== lib=${lib:-default}"
  GS_BENCH_LIB=${lib:+$R/$lib} timeout 600 python bench.py --no-cpu --no-other --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'blur_sobel_only', d['blur_sobel_only']['ms_per_step'], d['blur_sobel_only']['frac'], d['parity'][:60])"
done; done 2>&1 | tee gpurun_out/r04h_minw4_ab.log
echo "== ragged corner widths after the mode rule"; python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04h_corner_widths.log
import sys, os
sys.path.insert(0, os.getcwd())
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (w, h, n, off) in [(4094, 4096, 32, 0), (4095, 4096, 32, 0), (4096, 4096, 32, 1), (1023, 1024, 512, 0), (2047, 1024, 256, 0), (3839, 2160, 64, 0), (3840, 2160, 64, 3)]:
    sb = torch.randint(0, 256, (n * h * w + 64,), dtype=torch.uint8, device="cuda"); db = torch.zeros_like(sb)
    s = sb[off:off + n * h * w].view(n, h, w); d = db[off:off + n * h * w].view(n, h, w)
    row = []
    for mode in (0, 1, 2):
        g.tune(24, mode)
        row.append([timeit(lambda: g.sobel_batch(d, s)), timeit(lambda: g.blur_batch(d, s, 2)), timeit(lambda: g.erode_batch(d, s))])
    g.tune(24, 0)
    f = lambda ms: 2.0 * n * w * h / ms / 8e9
    print("%dx%d +%d  sobel/blur2/erode frac: rule %s | direct only %s | realign always %s" % (w, h, off, *[" ".join("%.3f" % f(x) for x in r) for r in row]), flush=True)
PY
