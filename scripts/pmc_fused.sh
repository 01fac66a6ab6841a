#!/bin/bash
# SQ counter passes (counters only, separate runs) over scripts/pmc_probe_fused.py
# (PMC_PROBE=<script> PMC_FILTER=<substring,substring> PMC_TAG=<dir prefix> for other kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
SETS=${PMC_SETS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU|SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"}
PROBE=${PMC_PROBE:-scripts/pmc_probe_fused.py}; export PMC_FILTER=${PMC_FILTER:-blur_sobel,strip_copy}; TAG=${PMC_TAG:-sqf}
IFS='|' read -ra A <<< "$SETS"
i=0
for c in "${A[@]}"; do
  i=$((i+1)); d=$R/gpurun_out/${TAG}_$i; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $R/$PROBE > /dev/null 2>$d.err
  python - "$d" <<PY
import csv,glob,sys,collections,os
FILT=os.environ["PMC_FILTER"].split(",")
hit=lambda k: any(f in k for f in FILT)
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no csv", sys.argv[1]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if hit(k): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-40s"%k[:40], {c: round(sum(x)/len(x)) for c,x in v.items()})
kt=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)
if kt:
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    print("  durations(us):", {k[:36]: round(sum(v)/len(v),1) for k,v in d.items() if hit(k)})
PY
done
