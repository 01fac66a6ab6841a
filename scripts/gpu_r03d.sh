#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== PMC, prefilter 2, 8 x 4K noise"; LBP_PRE=2 bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_lbp_pre2.txt
echo "== PMC, prefilter off"; LBP_PRE=-1 PMC_SETS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum|TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_lbp_off.txt
echo "== PMC, prefilter 2, ONE 4K frame (working set of one frame per XCD band)"; LBP_N=1 LBP_PRE=2 PMC_SETS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_lbp_pre2_n1.txt
