#!/usr/bin/env python3
"""what do launch boundaries and event records between the pipeline's fused launches cost?  512 4K frames:
(a) one gsh_blur_sobel_batch launch, (b) 16 launches of 32 frames back to back, (c) the same with an event
recorded after each launch (what gsh_edge_pipeline_batch does to release the side stream), (d) with two more
timing events per launch (what gsh_profile adds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
F, H, W = 512, 2160, 3840
src = torch.empty((F, H, W), dtype=torch.uint8, device="cuda"); g.synth_batch(src, 1000)
dst = torch.zeros_like(src)
st = torch.cuda.current_stream()
def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def one(): g.blur_sobel_batch(dst, src, 2)
def chunks(per, ev=0):
    evs = [torch.cuda.Event(enable_timing=(ev > 1)) for _ in range(3 * (F // per))]
    def run():
        k = 0
        for f0 in range(0, F, per):
            if ev > 1: evs[k].record(st); k += 1
            g.blur_sobel_batch(dst[f0:f0 + per], src[f0:f0 + per], 2)
            if ev > 1: evs[k].record(st); k += 1
            if ev: evs[k].record(st); k += 1
    return run
print("one launch of 512 frames          %.4f ms" % timeit(one))
for per in (32, 64, 128):
    print("%2d launches of %3d frames         %.4f ms" % (F // per, per, timeit(chunks(per))))
    print("   + 1 event record per launch    %.4f ms" % timeit(chunks(per, 1)))
    print("   + 3 event records per launch   %.4f ms" % timeit(chunks(per, 2)))
