#!/bin/bash
# round-2 late visit: counters for the feature kernels (-> profiles/fast_valu_pmc.json), gather microbench, then the full round script
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
PMC_PROBE=scripts/pmc_probe_features.py PMC_FILTER=k_fast,k_hist_partial,k_lbp,k_emit,k_chunk PMC_TAG=sqfeat \
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" bash scripts/pmc_fused.sh > gpurun_out/r02j_pmc_features.txt 2>&1
python scripts/pmc_fast_json.py > gpurun_out/pmc_fast_json.log 2>&1
timeout 200 build_variants/ubench_gather > gpurun_out/r02j_ubench_gather.log 2>&1
python scripts/ubench_hist.py > gpurun_out/r02j_hist.log 2>&1
python scripts/ubench_fast.py > gpurun_out/r02j_fast.log 2>&1
bash scripts/gpu_r02.sh
