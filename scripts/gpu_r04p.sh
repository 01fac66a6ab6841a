#!/bin/bash
# round 4, visit p: the any-radius box kernel with realigned loads on rows at byte phases that are no multiple of 4
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ragged box"; RG_CHECK=0 timeout 500 python scripts/ubench_ragged.py 2>&1 | grep -v amdgpu.ids | grep -E "^(blur r|adaptive|op )" | tee gpurun_out/r04p_ragged_box.log
timeout 900 python -m pytest tests/test_ragged.py tests/test_gpu_parity.py -q -m gpu -k "box or blur or adaptive" 2>&1 | tail -3 | tee gpurun_out/r04p_pytest.log
