#!/usr/bin/env python3
"""Generator (and verifier) of tests/golden/kat.json: every known-answer hash comes from the
UNMODIFIED reference compiled where it lies (oracle/_ref/libgs_ref.so, recipe oracle/Makefile) --
never from this repo's kernels or from the C restatement.  FNV-1a 32-bit over the raw output bytes,
inputs = the reference's lena.pgm fixture and the SURVEY.md 8(c) block-noise generator.

    python tests/golden/make_golden.py            # verify kat.json against the reference (exit 1 on a difference)
    python tests/golden/make_golden.py --write    # regenerate kat.json

Needs /root/reference (build container); the committed kat.json is what travels to the GPU box.
Takes about a minute (the 1080p cascade scan alone is ~6 s on one core)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from grayskull_amd.cascade import Cascade  # noqa: E402
from util import fnv, lena  # noqa: E402

SYNTH = [(3840, 2160, 1, False), (4096, 4096, 2, False), (1920, 1080, 3, True), (1280, 720, 4, True), (67, 45, 5, True)]
LBP = {"max_rects": 4096, "scale_factor": 1.1, "min_scale": 1.0, "max_scale": 4.0, "step": 1}
FAST = {"nkps": 5000, "threshold": 20}
ORB = {"nkps": 500, "threshold": 20}


def generate():
    if not pyoracle.have_reference():
        pyoracle.build()
    ref = Oracle("reference")
    casc = Cascade.from_blob(os.path.join(HERE, "frontalface_cascade.bin"))
    img = lena()
    ii = ref.integral(img)
    out = {"_comment": "Known-answer vectors produced by the UNMODIFIED reference (SURVEY.md 8c; regenerate/verify with "
                       "tests/golden/make_golden.py). FNV-1a 32-bit over raw output bytes. blur_sobel = gs_blur(r=2) then "
                       "gs_sobel into a zeroed dst; otsu/thr apply to that sobel output (for lena: to lena itself).",
           "lena": {"src": fnv(img), "blur": {str(r): fnv(ref.blur(img, r)) for r in (1, 2, 3, 9)},
                    "blur_sobel": fnv(ref.sobel(ref.blur(img, 2))), "otsu_src": int(ref.otsu_threshold(img)),
                    "thr_src": fnv(ref.threshold(img, ref.otsu_threshold(img))), "sobel": fnv(ref.sobel(img)),
                    "erode": fnv(ref.erode(img)), "dilate": fnv(ref.dilate(img)), "integral": fnv(ii),
                    "integral_last": int(ii[-1, -1]), "adaptive_r15_c5": fnv(ref.adaptive_threshold(img, 15, 5)),
                    "resize512_blur2_sobel": fnv(ref.sobel(ref.blur(ref.resize(img, 512, 512), 2)))},
           "synth": []}
    for (w, h, seed, feats) in SYNTH:
        s = Oracle.synth(w, h, seed)
        b = ref.blur(s, 2)
        e = ref.sobel(b)
        t = int(ref.otsu_threshold(e))
        k = {"w": w, "h": h, "seed": seed, "src": fnv(s), "blur2": fnv(b), "blur_sobel": fnv(e), "otsu": t,
             "thr": fnv(ref.threshold(e, t)), "sobel": fnv(ref.sobel(s)), "erode": fnv(ref.erode(s)),
             "dilate": fnv(ref.dilate(s)), "integral": fnv(ref.integral(s))}
        if feats:
            r = ref.lbp_detect(casc, ref.integral(s), LBP["max_rects"], LBP["scale_factor"], LBP["min_scale"],
                               LBP["max_scale"], LBP["step"])
            k["lbp"] = {"n": int(len(r)), "rects": fnv(r)}
            kp, sm = ref.fast(s, FAST["nkps"], FAST["threshold"])
            k["fast"] = {"n": int(len(kp)), "kp": fnv(kp), "scoremap": fnv(sm)}
            ok = ref.orb_extract(s, ORB["nkps"], ORB["threshold"])
            k["orb"] = {"n": int(len(ok)), "kp": fnv(ok)}
        out["synth"].append(k)
    out["lbp_params"], out["fast_params"], out["orb_params"] = LBP, FAST, ORB
    w, h, seed, (sx, sy) = 1280, 720, 4, (5, 3)
    A = Oracle.synth(w, h, seed)
    B = np.zeros_like(A)
    B[:h - sy, :w - sx] = A[sy:, sx:]
    m = {"w": w, "h": h, "seed": seed, "shift": [sx, sy], "nkps": 500, "threshold": 20, "max_matches": 2500,
         "max_distance": 60.0}
    mm = ref.match_orb(ref.orb_extract(A, 500, 20), ref.orb_extract(B, 500, 20), 2500, 60.0)
    m["n"], m["matches"] = int(len(mm)), fnv(mm)
    m["n_nkps2500"] = int(len(ref.match_orb(ref.orb_extract(A, 2500, 20), ref.orb_extract(B, 2500, 20), 2500, 60.0)))
    out["orb_match"] = m
    return out


def main():
    new = generate()
    path = os.path.join(HERE, "kat.json")
    if "--write" in sys.argv:
        json.dump(new, open(path, "w"), indent=1)
        open(path, "a").write("\n")
        print("wrote", path)
        return 0
    old = json.load(open(path))
    old.pop("_comment", None)
    new.pop("_comment", None)
    if old == new:
        print("kat.json matches the reference (%d synthetic frames, lena, ORB match)" % len(new["synth"]))
        return 0
    for key in sorted(set(old) | set(new)):
        if old.get(key) != new.get(key):
            print("DIFFERENT:", key, "\n  file:", old.get(key), "\n  reference:", new.get(key))
    return 1


if __name__ == "__main__":
    sys.exit(main())
