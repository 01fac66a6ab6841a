#!/bin/bash
# round 4, visit q: ragged rows on the register-ring box kernels that ignore the right edge + k_box_edge for the r rightmost columns
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ragged box (ring RAG + edge)"; RG_CHECK=0 timeout 500 python scripts/ubench_ragged.py 2>&1 | grep -v amdgpu.ids | grep -E "^(blur r|adaptive|op )" | tee gpurun_out/r04q_ragged_box.log
timeout 900 python -m pytest tests/test_ragged.py tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -q -m gpu -k "box or blur or adaptive" 2>&1 | tail -3 | tee gpurun_out/r04q_pytest.log
