#!/usr/bin/env python3
"""Generator (and verifier) of tests/golden/batch_checksums.json: per-frame results of BASELINE
configs[1] / configs[4] over the WHOLE 4096-frame batch, produced by the UNMODIFIED reference
(oracle/_ref/libgs_ref.so) -- never by this repo's kernels.  bench.py compares the per-frame
checksums every rank computes on its GPU with these, so that at any N every frame of every rank is
checked, not a sample of rank 0's.

configs[1], all frames f = 0..4095:  img = synth(3840, 2160, 1000 + f);  e = gs_sobel(gs_blur(img, 2))
into a zeroed dst;  t = gs_otsu_threshold(e);  gs_threshold(e, t).  Stored: t and
wsum(e) = sum_i (i + 1) * (byte_i + 1) mod 2^64 (what gsh_checksum_batch computes on the device).

configs[4], since round 6 every frame of the batch (`--add-cfg4 --all-cfg4`: 24 s of one host core per frame, the file is
rewritten every 32 frames; CFG4_FRAMES below is the subset a plain `--write` regenerates: one GPU's whole share plus a sample
of the rest):  e as above before thresholding;  ii = gs_integral(e);
gs_lbp_detect(frontalface, ii, 4096 rects, 1.1, 1.0, 4.0, step 1).  Stored: the count and wsum over the
count * 16 bytes of gs_rect records.

    python tests/golden/make_batch_golden.py --write [--jobs 8]     # ~35 min on 8 cores (24 s per configs[4] frame and core)
    python tests/golden/make_batch_golden.py [--sample 64]          # verify a random sample (exit 1 on a difference)

Needs /root/reference (build container); the committed JSON is what travels to the GPU box."""
import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

W, H, RADIUS, SEED0, FRAMES = 3840, 2160, 2, 1000, 4096
LBP = {"max_rects": 4096, "scale_factor": 1.1, "min_scale": 1.0, "max_scale": 4.0, "step": 1}
CFG4_BOUNDARY = sorted({0, 1, 2, 3} | {k * 512 for k in range(8)} | {k * 512 + 511 for k in range(8)})
# round 4: + 72 frames drawn once over the whole batch (fixed seed), so that every rank's shard at any N has frames to check
CFG4_RANDOM = sorted(int(x) for x in np.random.RandomState(4096).choice(4096, 72, replace=False))
# round 6: + every frame of the first 512 (the whole share of one GPU at N = 1 and of rank 0 at N = 8), so that the default
# bench run checks ALL of its configs[4] rect lists, not a sample
CFG4_FIRST_SHARE = list(range(512))
CFG4_FRAMES = sorted(set(CFG4_BOUNDARY) | set(CFG4_RANDOM) | set(CFG4_FIRST_SHARE))
OUT = os.path.join(HERE, "batch_checksums.json")


def wsum(a):
    """sum (i+1)*(byte+1) mod 2^64 over the raw bytes of `a` (gsh_checksum_batch, k_pointwise.h k_checksum)"""
    b = np.ascontiguousarray(a).view(np.uint8).reshape(-1).astype(np.uint64)
    return int(np.sum(np.arange(1, b.size + 1, dtype=np.uint64) * (b + np.uint64(1)), dtype=np.uint64))


_ref = None


def _oracle():
    global _ref
    if _ref is None:
        from oracle import pyoracle
        if not pyoracle.have_reference():
            pyoracle.build()
        _ref = pyoracle.Oracle("reference")
    return _ref


def cfg1_frame(f):
    from oracle.pyoracle import Oracle
    ref = _oracle()
    e = ref.sobel(ref.blur(Oracle.synth(W, H, SEED0 + f), RADIUS))
    t = int(ref.otsu_threshold(e))
    return f, t, wsum(ref.threshold(e, t))


def cfg4_frame(f):
    from oracle.pyoracle import Oracle
    from grayskull_amd.cascade import Cascade
    ref = _oracle()
    casc = Cascade.from_blob(os.path.join(HERE, "frontalface_cascade.bin"))
    e = ref.sobel(ref.blur(Oracle.synth(W, H, SEED0 + f), RADIUS))
    r = ref.lbp_detect(casc, ref.integral(e), LBP["max_rects"], LBP["scale_factor"], LBP["min_scale"],
                       LBP["max_scale"], LBP["step"])
    return f, int(len(r)), wsum(r)


def generate(jobs, frames1=None, frames4=None):
    frames1 = list(range(FRAMES)) if frames1 is None else frames1
    frames4 = CFG4_FRAMES if frames4 is None else frames4
    with mp.get_context("spawn").Pool(jobs) as pool:
        r4 = pool.map_async(cfg4_frame, frames4, chunksize=1)  # the long ones first
        r1 = pool.map(cfg1_frame, frames1, chunksize=8)
        r4 = r4.get()
    return r1, r4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--add-cfg4", action="store_true", help="generate the configs[4] frames of CFG4_FRAMES the file lacks and merge them in")
    ap.add_argument("--all-cfg4", action="store_true", help="with --add-cfg4: every frame of the batch, not only CFG4_FRAMES (~7 h on 4 cores)")
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--sample", type=int, default=32, help="verify mode: configs[1] frames to re-derive (0 = all)")
    ap.add_argument("--sample4", type=int, default=1, help="verify mode: configs[4] frames to re-derive")
    args = ap.parse_args()
    if args.write:
        r1, r4 = generate(args.jobs)
        out = {"_comment": "per-frame results of the UNMODIFIED reference over the 4096-frame batch of BASELINE configs[1] / "
                           "configs[4]; regenerate / verify with tests/golden/make_batch_golden.py.  wsum = sum (i+1)*(byte+1) "
                           "mod 2^64 over the raw output bytes, as a 16-digit hex string.",
               "w": W, "h": H, "radius": RADIUS, "seed0": SEED0, "frames": FRAMES,
               "cfg1": {"otsu": [t for _, t, _ in r1], "wsum": ["%016x" % s for _, _, s in r1]},
               "cfg4": {"params": LBP, "frames": {str(f): {"n": n, "wsum": "%016x" % s} for f, n, s in r4}}}
        with open(OUT, "w") as fh:
            json.dump(out, fh, separators=(",", ":"))
            fh.write("\n")
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
        return 0
    gold = json.load(open(OUT))
    if args.add_cfg4:
        want = list(range(FRAMES)) if args.all_cfg4 else CFG4_FRAMES
        missing = [f for f in want if str(f) not in gold["cfg4"]["frames"]]

        def save():
            gold["cfg4"]["frames"] = {k: gold["cfg4"]["frames"][k] for k in sorted(gold["cfg4"]["frames"], key=int)}
            with open(OUT + ".tmp", "w") as fh:
                json.dump(gold, fh, separators=(",", ":"))
                fh.write("\n")
            os.replace(OUT + ".tmp", OUT)

        done = 0
        with mp.get_context("spawn").Pool(args.jobs) as pool:  # 24 s per frame and core: the file is rewritten every 32 frames
            for f, n, s in pool.imap_unordered(cfg4_frame, missing, chunksize=1):
                gold["cfg4"]["frames"][str(f)] = {"n": n, "wsum": "%016x" % s}
                done += 1
                if done % 32 == 0:
                    save()
                    print("%d / %d" % (done, len(missing)), flush=True)
        save()
        print("added %d configs[4] frames" % done)
        return 0
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    f1 = list(range(FRAMES)) if args.sample == 0 else sorted(int(x) for x in rng.choice(FRAMES, args.sample, replace=False))
    have4 = [int(f) for f in gold["cfg4"]["frames"]]
    f4 = [int(x) for x in rng.choice(have4, min(args.sample4, len(have4)), replace=False)]
    r1, r4 = generate(args.jobs, f1, f4)
    bad = [f for f, t, s in r1 if gold["cfg1"]["otsu"][f] != t or gold["cfg1"]["wsum"][f] != "%016x" % s]
    bad += [("cfg4", f) for f, n, s in r4 if gold["cfg4"]["frames"][str(f)] != {"n": n, "wsum": "%016x" % s}]
    print("verified %d configs[1] frames, %d configs[4] frames against the reference: %s"
          % (len(r1), len(r4), "ok" if not bad else "MISMATCH at %s" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
