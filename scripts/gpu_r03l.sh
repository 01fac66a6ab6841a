#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
(echo "# 64 x 4096x4096, 256x1 blocks + XCD-aware bands; T = rows per band (0 = rule)"
UB_W=4096 UB_H=4096 UB_F=64 UB_T=0,2,3,4,5,6 UB_G=1 UB_PF=1 UB_OPS=sobel,blur2,erode,blur1 timeout 600 python scripts/ubench.py
echo "# 512 x 3840x2160"
UB_F=512 UB_T=0,3,4,5,6 UB_G=1 UB_PF=1 UB_OPS=sobel,blur2,erode timeout 600 python scripts/ubench.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ubench_T_xcd.log
