#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
PMC_PROBE=scripts/pmc_probe_tmatch.py PMC_FILTER=k_match_template,k_tm_ PMC_TAG=sqtm \
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT|SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_tmatch.txt
