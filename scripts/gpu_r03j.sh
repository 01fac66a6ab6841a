#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python scripts/bench_lbp_stages.py 1,2,3,4,5,6,7,8,9,10,12,14,17,20 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_stages_quad.log
