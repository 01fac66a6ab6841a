#!/bin/bash
# round 4, visit i: the C driver over real RCCL (world 1), the split library through the GPU suite's quick parts, the
# north-star launch in isolation under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== gsbatch over RCCL + dist + abi"; timeout 1200 python -m pytest tests/test_gsbatch.py tests/test_gpu_dist.py tests/test_abi.py -m gpu -q -x -p no:cacheprovider --durations=5 2>&1 | tail -12 | tee gpurun_out/r04i_pytest_rccl.log
echo "== rocprofv3: gs_sobel 64 x 4096^2 alone"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sobel4096 -o sobel4096 -- python $R/scripts/prof_sobel4096.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $R/gpurun_out/r04i_sobel4096.log
cd $R; f=$(find gpurun_out/prof_sobel4096 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04i_sobel4096_kernel_stats.csv && head -5 "$f" | cut -c1-200
