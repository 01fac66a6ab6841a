#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/ubench_fast_nms.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fast_nms.log
echo "== small frames, strip kernels"
(UB_W=1920 UB_H=1080 UB_F=8 UB_OPS=sobel,blur2,erode timeout 300 python scripts/ubench_xcd.py; UB_W=1920 UB_H=1080 UB_F=64 UB_OPS=sobel,blur2 timeout 300 python scripts/ubench_xcd.py; UB_W=1280 UB_H=720 UB_F=32 UB_OPS=sobel,blur2 timeout 300 python scripts/ubench_xcd.py; UB_W=640 UB_H=480 UB_F=64 UB_OPS=sobel,blur2 timeout 300 python scripts/ubench_xcd.py) 2>&1 | grep -v amdgpu.ids | grep "default rule" | tee gpurun_out/strip_small_frames.log
echo "== gpu tests (stencils, fast, pipeline)"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "not lbp and not cascade and not config4 and not dist" 2>&1 | tail -3
echo "== kernel times"; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast -o stats -- python $R/scripts/pmc_probe_features.py > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_fast -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "fast|emit|scan|Name|fill" "$f" | cut -c1-40,100-190 | tee gpurun_out/fast_kernel_stats.txt
