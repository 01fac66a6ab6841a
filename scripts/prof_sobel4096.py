#!/usr/bin/env python3
"""north-star shape in isolation (also the probe of rocprofv3 --kernel-trace --stats): gs_sobel alone on 64 distinct
4096 x 4096 frames (1 GiB per plane), 320 launches behind 3 untimed ones.  Every launch is bracketed by its own pair of HIP
events on the launch stream, so the log shows how the per-launch time moves while a FRESH process warms up -- the question
the round-4 review asked (0.598 of the HBM peak from a 23-launch process, 0.690 inside bench.py, same kernel, same frames) --
next to the shader / memory clocks rocm-smi reports before, in the middle of and after the run."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import grayskull_amd as gs


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.split(":", 1)[1].strip() if ":" in ln else ln for ln in out.splitlines() if "sclk" in ln or "mclk" in ln]
        return " | ".join(" ".join(k.split()) for k in keep[:4])
    except Exception as e:
        return "rocm-smi unavailable (%r)" % e


g = gs.lib(); g.use_torch_stream()
n, L = 64, int(os.environ.get("NS_LAUNCHES", 320))
a = torch.empty((n, 4096, 4096), dtype=torch.uint8, device="cuda"); g.synth_batch(a, 2)
b = torch.zeros_like(a)
print("clocks before:", clocks(), flush=True)
for _ in range(3): g.sobel_batch(b, a)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
mid = {}
th = threading.Thread(target=lambda: mid.update(c=clocks()))  # sampled while the launches below are in flight
ev[0].record()
for i in range(L):
    g.sobel_batch(b, a)
    ev[i + 1].record()
    if i == 20: th.start()
torch.cuda.synchronize(); th.join()
per = [ev[i].elapsed_time(ev[i + 1]) for i in range(L)]
by = float(n * (4096 * 4096 + 4094 * 4094))
frac = lambda ms: by / ms / 1e6 / 8000
avg = lambda v: sum(v) / len(v)
print("clocks during:", mid.get("c"))
print("clocks after:", clocks())
for name, v in (("launches 1-20", per[:20]), ("launches 21-100", per[20:100]), ("last 20", per[-20:]), ("all %d" % L, per)):
    print("%-16s %.4f ms per launch (min %.4f, max %.4f) = %.4f of 8 TB/s" % (name, avg(v), min(v), max(v), frac(avg(v))))
ms = ev[0].elapsed_time(ev[L]) / L
print("gs_sobel 64 x 4096x4096: %.4f ms per launch (HIP events over %d launches), %.1f GB/s algorithmic = %.4f of 8 TB/s, %.0f Mpix/s"
      % (ms, L, by / ms / 1e6, frac(ms), n * 4096 * 4096 / ms / 1e3))
