#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fast_q4.log
echo "== FAST/ORB gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fast or orb or keypoint or gsbatch or property or reference" 2>&1 | tail -3
