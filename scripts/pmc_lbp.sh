#!/bin/bash
# PMC passes (counters only, one set per run) over scripts/pmc_probe_lbp.py; summary per kernel -> stdout
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
SETS=${PMC_SETS:-"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum|SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY|TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS"}
IFS='|' read -ra A <<< "$SETS"
i=0
for c in "${A[@]}"; do
  i=$((i+1)); d=$R/gpurun_out/pmc_lbp_$i
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $R/scripts/pmc_probe_lbp.py > /dev/null 2>$d.err
  python - "$d" <<PY
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no csv", sys.argv[1]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); nl=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if k.startswith("gs::k_lbp"): acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(acc.items()): print("%-34s"%k[:34], {c: round(x) for c,x in v.items()}, "(summed over all launches)")
kt=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)
if kt:
    d=collections.defaultdict(float)
    for r in csv.DictReader(open(kt[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if k.startswith("gs::k_lbp"): d[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    print("  total durations(us):", {k[:34]: round(v,1) for k,v in d.items()})
PY
done
