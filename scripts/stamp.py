#!/usr/bin/env python3
"""Stamp a committed measurement file (profiles/*.json) with the hashes of the kernel sources it describes,
so that bench.py and the CPU tests can tell when the kernel changed after the measurement was taken:

    python scripts/stamp.py profiles/fused_isa_mix.json grayskull_amd/csrc/k_fused.h grayskull_amd/csrc/gs_fused.cpp

writes  "kernel_sources": {"<repo-relative path>": "<sha1 of the file>", ...}  and "stamped_at_commit"
(HEAD at the time) into the JSON.  bench.fresh() recomputes the hashes; a stale file's numbers are
reported as null, never silently."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha1(path):
    return hashlib.sha1(open(os.path.join(ROOT, path), "rb").read()).hexdigest()


def fresh(d, only=None):
    """True when every file named in d["kernel_sources"] still has the recorded hash (None: not stamped).
    only: check just these files of the stamp (a file with rows of several kernels; the caller names the sources of the
    kernel whose row it uses) -- a file the stamp does not name counts as changed"""
    ks = d.get("kernel_sources")
    if not ks:
        return None
    try:
        if only is not None:
            return all(p in ks and sha1(p) == ks[p] for p in only)
        return all(sha1(p) == h for p, h in ks.items())
    except OSError:
        return False


def main():
    path, files = sys.argv[1], sys.argv[2:]
    d = json.load(open(path))
    d["kernel_sources"] = {f: sha1(f) for f in files}
    try:
        d["stamped_at_commit"] = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %cs"], capture_output=True,
                                                text=True).stdout.strip()
    except Exception:
        pass
    json.dump(d, open(path, "w"), indent=1)
    print("stamped", path, "with", len(files), "source hashes")


if __name__ == "__main__":
    main()
