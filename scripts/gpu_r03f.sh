#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python scripts/bench_lbp_adaptive.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_adaptive.log
