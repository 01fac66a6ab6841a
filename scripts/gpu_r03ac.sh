#!/bin/bash
# SQ counters + FETCH/WRITE of the any-radius box kernel at r = 8 (gsh_tune key 6 = 4): the evidence behind k_box16r
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
BOX_ANY_RADIUS=1 PMC_PROBE=scripts/pmc_probe_box.py PMC_FILTER=k_box16 PMC_TAG=sqboxold \
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU|FETCH_SIZE|WRITE_SIZE" bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_box_any_radius.txt
