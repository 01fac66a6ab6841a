#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== hypothesis exploration on the GPU, 300 fresh random examples per property"
GS_HYPOTHESIS_EXAMPLES=300 timeout 1500 python -m pytest tests/test_property_shapes.py -m gpu -q -x --timeout 1400 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/property_explore_gpu.log
echo "== LBP property: prefilter on / quad off variants"
for k in "14 2" "17 1"; do set -- $k; GS_TUNE_KEY=$1 GS_TUNE_VAL=$2 GS_HYPOTHESIS_EXAMPLES=120 timeout 900 python -m pytest tests/test_property_shapes.py -m gpu -q -x -k lbp --timeout 800 -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/property_explore_gpu.log; done
