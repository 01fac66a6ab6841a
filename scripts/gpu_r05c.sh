#!/bin/bash
# round 5, third visit: k_lbp_tile v3 (own LDS tables without leaf values, truth tables in the dense phase too, add-with-carry
# code assembly, dense-to-pair switch sweep)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu -k lbp"; timeout 900 python -m pytest tests -m gpu -q -k "lbp or config4 or cfg4" --timeout 600 -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/r05c_pytest.log
echo "== bench_lbp_tile"; timeout 900 python scripts/bench_lbp_tile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c_lbp_tile.log
SETS="TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS|SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
echo "== counters, configs[4] edge maps, the rule"
LBP_EDGE=1 LBP_MODE=0 PMC_SETS="$SETS" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c_lbp_counters_tile_rule.txt
