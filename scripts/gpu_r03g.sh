#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
(UB_W=4096 UB_H=4096 UB_F=64 timeout 300 python scripts/ubench_xcd.py; UB_F=64 timeout 300 python scripts/ubench_xcd.py; UB_F=512 UB_OPS=sobel,blur2,erode timeout 300 python scripts/ubench_xcd.py; UB_W=1920 UB_H=1080 UB_F=8 UB_OPS=sobel,blur2 timeout 300 python scripts/ubench_xcd.py; UB_W=4096 UB_H=4096 UB_F=64 UB_T=4,16 UB_OPS=sobel,copy timeout 300 python scripts/ubench_xcd.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ubench_xcd.log
