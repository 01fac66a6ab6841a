#!/bin/bash
# round 4, visit j: counters of the LBP cascade's variants on the configs[4] input (8 x 4K edge maps, 2 calls each):
# default (quad-lane survivors), one lane per survivor window, and the row-sharing stage prefilter k_lbp_dense (2 stages)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
SETS="TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY|TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"
for v in "default:LBP_PRE=0" "one_lane:LBP_PRE=0 LBP_ONE_LANE=1" "prefilter2:LBP_PRE=2"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "== variant $tag ($envs)"
  env LBP_EDGE=1 $envs PMC_SETS="$SETS" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04j_lbp_counters_$tag.txt
done
