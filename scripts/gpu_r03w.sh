#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python scripts/ubench_tmatch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tmatch_mfma.log
echo "== geometry gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "template or geometry or reference or property" 2>&1 | tail -3
bash scripts/gpu_r03x.sh 2>&1 | tail -8
