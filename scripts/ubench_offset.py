#!/usr/bin/env python3
"""does the relative placement of src and dst in HBM matter for a two-buffer stream?
dst = one big allocation sliced at byte offset `off` past a 2 MiB-aligned base; src fixed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()  # UB_LIB=build_variants/libgs_experiment.so for the strip-copy probe
W, H, F = int(os.environ.get("UB_W", 3840)), int(os.environ.get("UB_H", 2160)), int(os.environ.get("UB_F", 64))
n = F * H * W
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
big = torch.empty(n + (64 << 20), dtype=torch.uint8, device="cuda")
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
offs = [0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, (1 << 20) + 4096, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, (1 << 24) + 65536 + 4096]
print("src %x big %x" % (src.data_ptr(), big.data_ptr()))
align = (-big.data_ptr()) % (1 << 21)
res = {}
for rnd in range(3):
    for off in offs:
        dst = big[align + off: align + off + n].view(F, H, W)
        for op, fn in (("copy", lambda: g.probe_strip_copy(dst, src)), ("sobel", lambda: g.sobel_batch(dst, src)),
                       ("erode", lambda: g.erode_batch(dst, src)), ("torch", lambda: dst.copy_(src))):
            res.setdefault((op, off), []).append(timeit(fn))
print("%-6s %10s %9s %8s" % ("op", "dst_off", "ms(med)", "GB/s"))
for (op, off), v in sorted(res.items()):
    ms = float(np.median(v)); print("%-6s %10d %9.4f %8.1f" % (op, off, ms, 2.0 * n / ms / 1e6))
