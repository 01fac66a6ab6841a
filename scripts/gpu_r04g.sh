#!/bin/bash
# round 4, visit g: the realigning strip flavour (RG 2) -- GPU parity of the ragged suite, byte-phase table with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest tests/test_ragged.py -m gpu"; timeout 900 python -m pytest tests/test_ragged.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/r04g_pytest_ragged.log
echo "== byte phase table (RG 2 on)"; timeout 300 python scripts/ubench_misalign.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04g_misalign_realign.log
echo "== byte phase table (key 24 = 1: direct loads)"; UB_TUNE24=1 timeout 300 python scripts/ubench_misalign.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04g_misalign_direct.log
echo "== ragged shapes"; RG_CHECK=0 RG_PART=1 timeout 300 python scripts/ubench_ragged.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04g_ragged.log | head -75
