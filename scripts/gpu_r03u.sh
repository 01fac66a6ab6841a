#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
PMC_PROBE=scripts/pmc_probe_fast.py PMC_FILTER=k_fast PMC_TAG=sqff \
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_SMEM|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_fast_fused.txt
