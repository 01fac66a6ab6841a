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
