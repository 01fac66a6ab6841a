#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp; rm -rf gpurun_out/prof_tm
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tm -o tm -- python $R/scripts/pmc_probe_tmatch.py > /dev/null 2>&1; cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_tm/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
for r in rows:
    n=r["Kernel_Name"].split("(")[0][-40:]
    if "tm_" in n or "match_template" in n or "sum_squares" in n:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        print("%-42s start +%8.1f us  dur %7.1f us  grid %s" % (n, 0 if t0 is None else (s-t0)/1e3, (e-s)/1e3, r.get("Grid_Size")))
        t0=s
PY
