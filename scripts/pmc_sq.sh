#!/bin/bash
# SQ/TA counter passes over scripts/pmc_probe.py (strip kernels + fused pipeline)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=$R/gpurun_out/sq_$(echo $c | cut -c1-14 | tr " " "_")
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python $R/scripts/pmc_probe.py > /dev/null 2>$d.err
  python - "$d" <<PY
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not f: print("no csv", sys.argv[1]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if k.startswith("gs::k_") and ("16" in k or "strip" in k or "threshold" in k or "hist_partial" in k): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print("%-34s"%k[:34], {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
