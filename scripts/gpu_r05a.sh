#!/bin/bash
# round 5, first visit: k_lbp_tile parity on hardware + per-shape / per-scale timings against k_lbp_cascade
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu -k lbp"; timeout 900 python -m pytest tests -m gpu -q -k "lbp or config4 or cfg4" --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r05a_pytest_lbp.log
echo "== bench_lbp_tile"; timeout 900 python scripts/bench_lbp_tile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05a_lbp_tile.log
