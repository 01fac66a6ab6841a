#!/usr/bin/env python3
"""round 4: the strip kernels on ragged widths (w % 16 != 0) and frames at odd byte addresses (run on the GPU box).
Part 1: bit-exact against the oracle inside sentinel-guarded buffers.  Part 2: time per launch, aligned vs base+1 vs
ragged vs the round-3 per-pixel kernels (gsh_tune key 21 = 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import grayskull_amd as gs
from oracle.pyoracle import Oracle

g = gs.Grayskull(os.environ["UB_LIB"]) if os.environ.get("UB_LIB") else gs.lib(); g.use_torch_stream()
o = Oracle("port")
rs = np.random.RandomState(4)
GUARD = 256

def guarded(n, h, w, off, fill):
    buf = torch.full((GUARD + off + n * h * w + GUARD,), 0xAB, dtype=torch.uint8, device="cuda")
    v = buf[GUARD + off:GUARD + off + n * h * w].view(n, h, w)
    if fill is not None: v.copy_(torch.from_numpy(fill))
    return buf, v

def guards_ok(buf, off, nbytes):
    a = buf[:GUARD + off].cpu().numpy(); b = buf[GUARD + off + nbytes:].cpu().numpy()
    return bool((a == 0xAB).all() and (b == 0xAB).all())

bad = 0
k3 = np.array([[1, -2, 1], [2, 4, -2], [1, 2, 1]], np.int8)
if os.environ.get("RG_CHECK", "1") == "1":
    for w in [32, 33, 47, 63, 65, 100, 612, 1009, 1023, 1041, 1080, 2065, 3838]:
        for off in (0, 1, 7):
            h, n = 23, 3
            img = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
            d0 = rs.randint(0, 256, (n, h, w)).astype(np.uint8)
            sb, s = guarded(n, h, w, off, img)
            exp = {"sobel": np.stack([o.sobel(img[i], d0[i]) for i in range(n)])}
            for r in (1, 2, 3): exp["blur%d" % r] = np.stack([o.blur(img[i], r) for i in range(n)])
            exp["erode"] = np.stack([o.erode(img[i]) for i in range(n)]); exp["dilate"] = np.stack([o.dilate(img[i]) for i in range(n)])
            exp["filter"] = np.stack([o.filter(img[i], k3, 8) for i in range(n)])
            for r in (5, 11): exp["blur%d" % r] = np.stack([o.blur(img[i], r) for i in range(n)])  # ring kernel + k_box_edge on ragged rows
            exp["adaptive7"] = np.stack([o.adaptive_threshold(img[i], 7, 5) for i in range(n)])
            for name in exp:
                db, d = guarded(n, h, w, off, d0)
                if name == "sobel": g.sobel_batch(d, s)
                elif name.startswith("blur"): g.blur_batch(d, s, int(name[4:]))
                elif name == "adaptive7": g.adaptive_threshold_batch(d, s, 7, 5)
                elif name == "erode": g.erode_batch(d, s)
                elif name == "dilate": g.dilate_batch(d, s)
                else: g.filter_batch(d, s, k3, 8)
                torch.cuda.synchronize()
                got = d.cpu().numpy()
                if not np.array_equal(got, exp[name]) or not guards_ok(db, off, n * h * w):
                    bad += 1
                    idx = np.argwhere(got != exp[name])
                    print("MISMATCH", name, w, off, len(idx), idx[:3].tolist(), "guards", guards_ok(db, off, n * h * w))
            # single-frame FAST through the drop-in call on a device view
            sm0 = rs.randint(0, 256, (h, w)).astype(np.uint8)
            smb, sm = guarded(1, h, w, off, sm0[None])
            kp = g.fast(s[0], sm[0], 5000, 20)
            ko, smo = o.fast(img[0], 5000, 20, sm0)
            if not np.array_equal(kp, ko) or not np.array_equal(sm[0].cpu().numpy(), smo) or not guards_ok(smb, off, h * w):
                bad += 1; print("MISMATCH fast", w, off)
        print("check w=%d ok" % w, flush=True)
    print("PARITY", "green" if bad == 0 else "RED %d" % bad)

def timeit(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

F = int(os.environ.get("RG_F", 64))
shapes = [(3840, 2160, 0), (3840, 2160, 1), (3838, 2160, 0), (3838, 2160, 5), (4096, 4096, 0), (4094, 4096, 0), (1920, 1080, 0), (1080, 1920, 0), (1080, 1920, 3), (612, 816, 0), (1366, 768, 0)]
ops = {"copy": lambda d, s: g.probe_strip_copy(d, s), "sobel": lambda d, s: g.sobel_batch(d, s), "blur2": lambda d, s: g.blur_batch(d, s, 2),
       "blur1": lambda d, s: g.blur_batch(d, s, 1), "erode": lambda d, s: g.erode_batch(d, s), "filter": lambda d, s: g.filter_batch(d, s, k3, 8)}
print("%-7s %5s %5s %3s %3s %9s %8s %6s   %s" % ("op", "w", "h", "off", "F", "ms", "GB/s", "frac", "old-rule ms (per-pixel kernels)"))
for (w, h, off) in shapes:
    n = F if w * h <= 3840 * 2160 else max(8, F // 2)
    if w * h < 1920 * 1080: n = F * 4
    sb, s = guarded(n, h, w, off, None); g.synth_batch(s, 1000) if (off == 0 and w % 16 == 0) else s.copy_(torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda"))
    db, d = guarded(n, h, w, off, None)
    for name, fn in ops.items():
        ms = timeit(lambda: fn(d, s))
        old = ""
        if (w % 16 or off) and name != "copy":
            g.tune(21, 1); old = "%.4f" % timeit(lambda: fn(d, s), 3); g.tune(21, 0)
        gbs = 2.0 * n * w * h / ms / 1e6
        print("%-7s %5d %5d %3d %3d %9.4f %8.1f %6.3f   %s" % (name, w, h, off, n, ms, gbs, gbs / 8000, old), flush=True)

if os.environ.get("RG_PART") == "1": sys.exit(0)
# ---- round 4, second part: gs_integral (banded form), the sliding box, gs_downsample on the same shapes
print("%-14s %5s %5s %3s %3s %9s %8s %6s   %s" % ("op", "w", "h", "off", "F", "ms", "GB/s(alg)", "frac", "old-rule ms"))
for (w, h, off, n) in [(3840, 2160, 0, 64), (3838, 2160, 0, 64), (612, 816, 0, 64), (612, 816, 0, 256), (1080, 1920, 0, 64), (1920, 1080, 0, 64), (7680, 4320, 0, 8)]:
    sb, s = guarded(n, h, w, off, None); s.copy_(torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda"))
    db, d = guarded(n, h, w, off, None)
    ii = torch.empty((n, h, w), dtype=torch.int32, device="cuda")
    half = torch.empty((n, h // 2, w // 2), dtype=torch.uint8, device="cuda")
    more = {"integral": (lambda: g.integral_batch(s, ii), 5.0), "blur r=5": (lambda: g.blur_batch(d, s, 5), 2.0), "blur r=16": (lambda: g.blur_batch(d, s, 16), 2.0),
            "adaptive r=15": (lambda: g.adaptive_threshold_batch(d, s, 15, 5), 2.0), "downsample": (lambda: g.downsample_batch(half, s), 1.25)}
    for name, (fn, bpp) in more.items():
        if w > 4096 and name != "integral" and name != "downsample": continue
        ms = timeit(fn)
        old = ""
        if (w % 16 or off or w > 4096):
            g.tune(21, 1); old = "%.4f" % timeit(fn, 3); g.tune(21, 0)
        gbs = bpp * n * w * h / ms / 1e6
        print("%-14s %5d %5d %3d %3d %9.4f %8.1f %6.3f   %s" % (name, w, h, off, n, ms, gbs, gbs / 8000, old), flush=True)
    del ii, half, sb, db
