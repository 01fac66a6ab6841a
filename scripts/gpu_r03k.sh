#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
UB_W=1920 UB_H=1080 UB_F=8 timeout 900 python scripts/bench_lbp_adaptive.py "8,2,1,3,6;8,2,2,5,0;8,2,2,4,8;8,2,1,2,4" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lbp_adaptive_1080p.log
UB_W=1920 UB_H=1080 UB_F=1 timeout 900 python scripts/bench_lbp_adaptive.py "8,2,1,3,6;8,2,2,5,0;8,2,2,4,8;8,2,1,2,4" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/lbp_adaptive_1080p.log
done
