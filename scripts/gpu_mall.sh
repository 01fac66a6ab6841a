#!/bin/bash
# k_threshold durations, warm vs cold, by frames (see scripts/ubench_mall.py); the launch order is fixed: per n, 4 x (warm, cold)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for t in "" st0; do
  d=$R/gpurun_out/mall_$t; rm -rf $d
  UB_TAG=$t timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o m -- python $R/scripts/ubench_mall.py > /dev/null 2>$d.err
  python - "$d" "$t" <<PY
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_threshold" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("# library:", sys.argv[2] or "this tree (nt stores)")
i=0
for n in (2,4,8,16,32):
    w=[d[i+2*k] for k in range(4)]; c=[d[i+2*k+1] for k in range(4)]; i+=8
    mb=n*3840*2160/1e6
    print("%2d frames (%4.0f MB): k_threshold right after k_sobel wrote them %6.1f us (%.2f TB/s R+W)   after 600 MB of other traffic %6.1f us (%.2f TB/s)"%(n,mb,min(w),2*mb/min(w),min(c),2*mb/min(c)))
PY
done
