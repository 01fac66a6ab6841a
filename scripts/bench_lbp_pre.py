#!/usr/bin/env python3
"""LBP cascade with the stage prefilter k_lbp_dense (gsh_tune key 14: 0 = off, k = prefiltered stages; default 2) on
block-noise frames (configs[2]) and edge maps (configs[4]); identical rectangles checked by checksum.
usage: bench_lbp_pre.py [keys: comma list, default -1,1,2,3,4]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
from grayskull_amd.cascade import Cascade
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
keys = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4").split(",")]
k17s = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]  # 1 = one lane per re-packed window (round 2)
g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
casc = Cascade.from_blob(os.path.join(ROOT, "tests/golden/frontalface_cascade.bin"))
dc = g.cascade_create(casc)
def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (w, h, n) in ((1920, 1080, 8), (3840, 2160, 8), (1920, 1080, 1)):
    src = torch.empty((n, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000 if n > 1 else 3)
    a, b = torch.empty_like(src), torch.zeros_like(src)
    g.blur_batch(a, src, 2); g.sobel_batch(b, a)
    for name, img in (("noise", src), ("edges", b)):
        ii = torch.zeros((n, h, w), dtype=torch.int32, device="cuda"); g.integral_batch(img, ii)
        rects = torch.zeros((n, 4096, 4), dtype=torch.int32, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
        for key in [(k, q) for k in keys for q in k17s]:
            key, k17 = key
            g.tune(14, key); g.tune(17, k17)
            ms = timeit(lambda: g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1))
            crc = zlib.crc32(rects.cpu().numpy().tobytes()) ^ zlib.crc32(counts.cpu().numpy().tobytes())
            ev = torch.zeros(4, dtype=torch.int64, device="cuda"); g.lbp_count_evaluated(ev)
            g.lbp_detect_batch(dc, ii, rects, counts, 4096, 1.1, 1.0, 4.0, 1); torch.cuda.synchronize(); g.lbp_count_evaluated(None)
            e = ev.cpu().tolist()
            print("%s %dx%d x%d  prefilter %2d quad %d  %.3f ms/frame  counts %s crc %08x  windows %d weak/win %.2f loads/win %.1f pre-windows %d"
                  % (name, w, h, n, key, 1 - k17, ms / n, counts.cpu().tolist()[:2], crc, e[0] // n, e[1] / max(e[0], 1), e[2] / max(e[0], 1), e[3] // n), flush=True)
        g.tune(14, 0); g.tune(17, 0)
dc.close()
