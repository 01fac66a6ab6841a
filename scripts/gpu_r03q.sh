#!/bin/bash
# k_fast_score_q4 with XCD-aware tile order (default) vs launch order (key 18 = 1); k_emit with its loads issued together
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 0 1; do echo "== key 18 = $k"; UB_K18=$k UB_ONLY=0 timeout 600 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/fast_xcd.log
echo "== FAST/ORB gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fast or orb or keypoint or gsbatch or property or reference or match" 2>&1 | tail -3
echo "== PMC FETCH/WRITE of the FAST kernels"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcq_$c -o pmc -- python $R/scripts/pmc_probe_fast.py > $R/gpurun_out/pmcq_$c.log 2>&1
  cd $R; python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcq_$c/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "fast" in k or "emit" in k: print("$c KB (FETCH x 2 = bytes read)", k, [round(x) for x in v], "(first half: XCD tile order, second half: launch order)")
PY
done 2>&1 | tee gpurun_out/fast_xcd_pmc.log
