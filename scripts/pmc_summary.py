#!/usr/bin/env python3
"""per-kernel mean FETCH_SIZE / WRITE_SIZE from the rocprofv3 --pmc csv outputs"""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(c, ": no counter_collection csv found"); continue
    acc = collections.defaultdict(list)
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == c:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    print("== %s (raw counter units as reported by rocprofv3; guide: KB for *_SIZE)" % c)
    for k, v in sorted(acc.items()):
        print("  %-60s n=%3d mean=%.1f" % (k[:60], len(v), sum(v) / len(v)))

# machine-readable per-launch HBM traffic for bench.py (gfx950 corrections per MI355X_MICROARCH.md:
# FETCH_SIZE is reported in KB and counts HALF of wide coalesced reads -> x2; WRITE_SIZE in KB, exact;
# both confirmed on k_strip_copy, whose byte count is known)
import json
traffic = {}
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    if not files: continue
    acc = collections.defaultdict(list)
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == c:  # launches of one kernel with different grids are different workloads
                acc[(row["Kernel_Name"].split("(")[0].replace("void ", ""), row.get("Grid_Size", "?"))].append(float(row["Counter_Value"]))
    for k, v in acc.items(): vals.setdefault(k, {})[c] = sum(v) / len(v)
ngrids = collections.Counter(k for k, _ in vals)
for (k, grid), v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k.startswith("gs::"):
        traffic[k if ngrids[k] == 1 else "%s@grid%s" % (k, grid)] = {
            "read_bytes": round(2 * v["FETCH_SIZE"] * 1024), "write_bytes": round(v["WRITE_SIZE"] * 1024),
            "total_bytes": round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "grid_threads": grid}
out = {"workload": "64 frames 3840x2160 (scripts/pmc_probe.py); the pipeline call launches its fused kernel per 32-frame chunk",
       "frames_per_launch_default": 64, "fused_frames_per_launch": 32,  # the pipeline launches k_blur_sobel_hist16 per 32-frame chunk
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; read = 2 x FETCH_SIZE KB (gfx950), write = WRITE_SIZE KB",
       "per_launch": traffic}
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
print("wrote", os.path.join(root, "pmc_traffic.json"))
