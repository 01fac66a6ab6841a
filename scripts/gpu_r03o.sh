#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/bench_lbp_pre.py 0 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_now2.log
echo "== LBP gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "lbp or cascade or config4" 2>&1 | tail -3
