#!/bin/bash
# round 5, second visit: k_lbp_tile with occupancy-grouped launches, stage truth tables, more shapes; counters of both kernels;
# the GS_NO_STDLIB seam through the C ABI
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu -k 'lbp or nostdlib or config4'"; timeout 900 python -m pytest tests -m gpu -q -k "lbp or config4 or cfg4 or nostdlib or no_stdlib" --timeout 600 -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r05b_pytest.log
echo "== bench_lbp_tile"; timeout 900 python scripts/bench_lbp_tile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b_lbp_tile.log
SETS="TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS|SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES|TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
for mode in 1 0; do
  echo "== counters, configs[4] edge maps, key 14 = $mode"
  LBP_EDGE=1 LBP_MODE=$mode PMC_SETS="$SETS" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b_lbp_counters_mode$mode.txt
done
