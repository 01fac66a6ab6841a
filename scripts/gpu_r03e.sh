#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== quad-lane survivors A/B"; timeout 600 python scripts/bench_lbp_pre.py -1,2 1,0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_quad.log
echo "== LBP gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "lbp or cascade or config4" 2>&1 | tail -3
