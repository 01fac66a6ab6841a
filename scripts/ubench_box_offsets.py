#!/usr/bin/env python3
"""round 3 left one discrepancy open: gs_blur(r = 16) on 64 x 4K took 0.27 ms in one script and 0.33 ms in another on the same
box -- same call, same data, buffers allocated in a different order.  This script holds everything else fixed and moves
only WHERE the two planes lie: both inside one 2 GiB arena, destination at a swept offset behind the source."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
F, H, W = 64, 2160, 3840
nb = F * H * W
arena = torch.empty(3 * nb + (64 << 20), dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
pad = (-base) % (2 << 20)            # source at a 2 MiB boundary
src = arena[pad:pad + nb].view(F, H, W); g.synth_batch(src, 1000)
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
first = pad + ((nb + (2 << 20) - 1) // (2 << 20)) * (2 << 20)   # first 2 MiB boundary behind the source
print("src at 2 MiB boundary; dst = first 2 MiB boundary behind it + offset")
print("%12s %10s %10s %10s %10s" % ("offset", "blur r=16", "blur r=9", "adapt r=15", "sobel"))
for off in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 65536, 8 << 20, (8 << 20) + 256, 32 << 20):
    d = arena[first + off:first + off + nb].view(F, H, W)
    print("%12d %10.4f %10.4f %10.4f %10.4f" % (off, timeit(lambda: g.blur_batch(d, src, 16)), timeit(lambda: g.blur_batch(d, src, 9)),
                                             timeit(lambda: g.adaptive_threshold_batch(d, src, 15, 5)), timeit(lambda: g.sobel_batch(d, src))), flush=True)
# and the two allocation orders of round 3's scripts, torch's allocator deciding
for order in ("src first", "dst first"):
    torch.cuda.empty_cache()
    if order == "src first":
        s2 = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); d2 = torch.empty_like(s2)
    else:
        d2 = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); s2 = torch.empty_like(d2)
    g.synth_batch(s2, 1000)
    print("%-10s src %#x dst %#x (dst - src = %d): blur r=16 %.4f ms" % (order, s2.data_ptr(), d2.data_ptr(), d2.data_ptr() - s2.data_ptr(), timeit(lambda: g.blur_batch(d2, s2, 16))))
    del s2, d2
