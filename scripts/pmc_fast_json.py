#!/usr/bin/env python3
"""profiles/fast_valu_pmc.json from the SQ_INSTS_VALU pass of scripts/pmc_probe_features.py (gpurun_out/sqfeat_1):
VALU wave-instructions per pixel of the gs_fast score kernels on the configs[3] block-noise frames.
avg_issue_cycles: issue-class mix of the executed blocks of k_fast_score_tile in hipcc's assembly (compass block +
ring block + the two run tests: 48 plain VOP1/VOP2 at 2 cycles, 114 VOP3 / cmp / select at 4), `make -C grayskull_amd/csrc asm`."""
import csv, glob, json, os, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "sqfeat_1")
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if r["Counter_Name"] == "SQ_INSTS_VALU" and "k_fast_score" in k:
        acc[k].append(float(r["Counter_Value"]))
npx = 32 * 1280 * 720
out = {"workload": "32 x 1280x720 synth(seed 4), threshold 20 (scripts/pmc_probe_features.py)", "avg_issue_cycles": round((48 * 2 + 114 * 4) / 162.0, 3),
       "source": "rocprofv3 --pmc SQ_INSTS_VALU (profiles/r02l_pmc_features.txt); issue cycles from profiles/r02a_ubench_valu.log classes",
       "kernels": {k: {"valu_wave_insts_per_launch": sum(v) / len(v), "valu_wave_insts_per_px": sum(v) / len(v) / npx} for k, v in acc.items()}}
tile = [v for k, v in out["kernels"].items() if "score_q4" in k] or [v for k, v in out["kernels"].items() if "tile" in k]
out["valu_wave_insts_per_px"] = tile[0]["valu_wave_insts_per_px"] if tile else None
for p in (os.path.join(root, "gpurun_out", "fast_valu_pmc.json"), os.path.join(root, "profiles", "fast_valu_pmc.json")):
    json.dump(out, open(p, "w"), indent=1)
print(json.dumps(out))
