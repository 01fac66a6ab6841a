#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/ubench_next_rows.py 2>&1 | grep -v amdgpu.ids | grep "box\|adaptive"
UB_BIG_ONLY=1 timeout 300 python scripts/ubench_box_ring.py 2>&1 | grep -v amdgpu.ids | grep "gs_blur  *r=16\|gs_blur  *r=5 \|adaptive_threshold  *r=8 \|adaptive_threshold  *r=2 "
timeout 300 python scripts/ubench_next_rows.py 2>&1 | grep -v amdgpu.ids | grep "box\|adaptive"
