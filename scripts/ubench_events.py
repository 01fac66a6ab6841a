#!/usr/bin/env python3
"""config-2 pipeline step time vs the flags of the events that order the library's two streams:
this tree (hipEventDisableSystemFence) vs HIP's default events vs hipEventReleaseToDevice (build_variants/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = os.environ.get("UB_LIB_TAG", "this tree")
libs = {name: gs.lib() if name == "this tree" else gs.Grayskull(os.path.join(ROOT, "build_variants", "libgs_%s.so" % name))}
F, H, W = 512, 2160, 3840
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); libs[name].use_torch_stream(); libs[name].synth_batch(src, 1000)
dst = torch.zeros_like(src)
hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda"); thr = torch.zeros(F, dtype=torch.uint8, device="cuda")
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for rnd in range(3):
    for name, g in libs.items():
        g.use_torch_stream()
        for prof in (0, 1):
            g.profile(400 if prof else 0)
            ms = timeit(lambda: g.edge_pipeline_batch(dst, None, src, 2, hist, thr))
            g.profile_read(); g.profile(0)
            cs = (int(dst.view(torch.int32).sum().item()) & 0xffffffff, int(thr.sum()))
            ref = ref or cs
            print("%-24s profile events %d  %.4f ms  %.0f Mpix/s %s" % (name, prof, ms, F * W * H / ms / 1e3, "ok" if cs == ref else "MISMATCH"))
