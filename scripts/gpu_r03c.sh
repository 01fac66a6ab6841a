#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fixed cost"; timeout 300 python scripts/bench_lbp_fixed.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_fixed.log
echo "== stages"; timeout 900 python scripts/bench_lbp_stages.py 1,2,3,4,5,6,7,8,9,10,12,14,17,20 -1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_stages.log
