#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/ubench_box_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/box_ring.log
echo "== blur / adaptive / property gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "blur or adaptive or next_rows or property or reference or stencil or box" 2>&1 | tail -3
