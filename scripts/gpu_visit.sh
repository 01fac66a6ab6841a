#!/bin/bash
# One GPU-box visit, parametrised: `gpurun -- 'bash scripts/gpu_visit.sh TAG section [section ...]'`.
# Logs go to gpurun_out/<TAG>_<section>.*; copy what is to be judged into profiles/ afterwards.
# Sections (rounds 2-4 had one script per visit, ~50 of them; this is their union):
#   smoke        __graft_entry__.smoke()
#   suite        pytest -m gpu (everything);  suite:EXPR  pytest -m gpu -k EXPR
#   bench        python bench.py (the driver's default line)
#   cfg4         bench.py --workload cfg4 --frames 128 (configs[4], golden-checked)
#   strong       bench.py --scaling strong --frames 4096 (the whole batch on this GPU: the denominator of north_star's ratio)
#   rccl1        bench.py under torch.distributed.run, one rank, GS_BENCH_FORCE_DIST=1 (nccl == RCCL at world 1)
#   rehearsal    bench.py --gpus 8 over gloo on this one GPU: weak, strong, cfg4 with ranks that own no frame
#   prof         rocprofv3 --kernel-trace --stats of the bench command
#   pmc          FETCH_SIZE / WRITE_SIZE passes over scripts/pmc_probe.py + summary (profiles/pmc_traffic.json workflow)
#   sobel4096    the north-star launch in a fresh process: 320 launches with per-launch events and clocks, alone and under rocprofv3
#   lbp          scripts/bench_lbp_tile.py (every tile shape, the rule, per scale);  lbp:quick  whole scans only
#   lbpstages    scripts/bench_lbp_stages.py on the experiment library (needs build_variants/libgs_experiment.so)
#   lbppmc       counters of the LBP kernels on the configs[4] input: the rule and k_lbp_cascade
#   fast         scripts/ubench_fast.py (flat / block noise / lena / random)
#   fastpmc      SQ_INSTS_VALU of the gs_fast passes -> profiles/fast_valu_pmc.json workflow
#   extras       ragged shapes, box offsets, next rows, template matching micro-benchmarks
#   boxragged    scripts/ubench_box_ragged.py (sliding box on ragged / aligned batches, k_box_edge on the side / the caller's stream) + its kernels under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:?usage: gpu_visit.sh TAG section...}; shift
O=gpurun_out/$TAG
for sec in "$@"; do
  arg=${sec#*:}; [ "$arg" = "$sec" ] && arg=""
  echo "== $sec"
  case ${sec%%:*} in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee ${O}_smoke.log ;;
    suite) timeout 2400 python -m pytest tests -m gpu -q ${arg:+-k "$arg"} --timeout 900 -p no:cacheprovider --durations=5 2>&1 | tail -12 | tee ${O}_pytest_gpu.log ;;
    bench) timeout 900 python bench.py 2>${O}_bench.err | tee ${O}_bench.json | cut -c1-600; tail -3 ${O}_bench.err ;;
    cfg4) timeout 900 python bench.py --workload cfg4 --frames 128 --steps 1 --warmup 1 2>${O}_cfg4.err | tee ${O}_bench_cfg4.json | cut -c1-1200 ;;
    strong) timeout 900 python bench.py --scaling strong --frames 4096 --steps 10 --warmup 2 --no-cpu --no-other 2>${O}_strong.err | tee ${O}_bench_strong4096_n1.json | cut -c1-500 ;;
    rccl1) GS_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --no-other --no-cpu --steps 20 2>${O}_rccl1.err | grep '^{' | tee ${O}_bench_rccl_world1.json | cut -c1-400 ;;
    rehearsal)
      for mode in weak strong; do
        GS_BENCH_BACKEND=gloo GS_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling $mode --frames $([ $mode = weak ] && echo 8 || echo 64) --steps 3 --warmup 1 --no-cpu --no-other 2>${O}_reh_$mode.err | grep '^{' | tee ${O}_bench_8rank_rehearsal_gloo_$mode.json | cut -c1-700
        tail -2 ${O}_reh_$mode.err
      done
      GS_BENCH_BACKEND=gloo GS_BENCH_DEVICE=0 timeout 900 python bench.py --gpus 8 --scaling strong --workload cfg4 --frames 5 --steps 1 --warmup 1 2>${O}_reh_cfg4.err | grep '^{' | tee ${O}_bench_8rank_rehearsal_gloo_cfg4.json | cut -c1-1500
      tail -2 ${O}_reh_cfg4.err ;;
    prof)
      rm -rf gpurun_out/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --no-cpu --no-verify > $R/${O}_prof_bench.json 2> $R/${O}_prof.err)
      f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" ${O}_bench_kernel_stats.csv && head -30 "$f" | cut -c1-150 ;;
    pmc)
      rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/scripts/pmc_probe.py > $R/gpurun_out/pmc_$c.log 2>&1)
      done
      python scripts/pmc_summary.py gpurun_out 2>&1 | tee ${O}_pmc_summary.txt | tail -40 ;;
    sobel4096)
      timeout 600 python scripts/prof_sobel4096.py 2>&1 | grep -v amdgpu.ids | tee ${O}_sobel4096.log
      rm -rf gpurun_out/prof_sobel4096
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sobel4096 -o sobel4096 -- python $R/scripts/prof_sobel4096.py 2>&1 | grep "gs_sobel 64" | tee $R/${O}_sobel4096_under_rocprof.log)
      f=$(find gpurun_out/prof_sobel4096 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" ${O}_sobel4096_kernel_stats.csv && head -4 "$f" | cut -c1-220 ;;
    lbp) timeout 900 python scripts/bench_lbp_tile.py $arg 2>&1 | grep -v amdgpu.ids | tee ${O}_lbp_tile.log ;;
    lbpstages) UB_LIB=$R/build_variants/libgs_experiment.so timeout 900 python scripts/bench_lbp_stages.py 2>&1 | grep -v amdgpu.ids | tee ${O}_lbp_stage_costs.log ;;
    lbppmc)
      SETS="TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS|SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY|SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES|TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
      for mode in 0 1; do
        LBP_EDGE=1 LBP_MODE=$mode PMC_SETS="$SETS" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee ${O}_lbp_counters_mode$mode.txt
      done ;;
    fast) timeout 600 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | tee ${O}_fast.log ;;
    fastpmc)
      PMC_PROBE=scripts/pmc_probe_features.py PMC_FILTER=k_fast,k_emit PMC_TAG=sqfeat \
        PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee ${O}_pmc_features.txt
      python scripts/pmc_fast_json.py > ${O}_pmc_fast_json.log 2>&1; tail -2 ${O}_pmc_fast_json.log | cut -c1-300 ;;
    extras)
      export UB_LIB=$R/build_variants/libgs_experiment.so  # the strip-copy probe lives in experiment builds only
      RG_CHECK=0 timeout 500 python scripts/ubench_ragged.py 2>&1 | grep -v amdgpu.ids | tee ${O}_ragged.log | tail -40
      timeout 300 python scripts/ubench_box_offsets.py 2>&1 | grep -v amdgpu.ids | tee ${O}_box_offsets.log
      timeout 300 python scripts/ubench_next_rows.py 2>&1 | grep -v amdgpu.ids | tee ${O}_next_rows.log | tail -12
      timeout 300 python scripts/ubench_tmatch.py 2>&1 | grep -v amdgpu.ids | tee ${O}_tmatch.log | tail -12
      unset UB_LIB ;;
    boxragged)
      timeout 600 python scripts/ubench_box_ragged.py 2>&1 | grep -v amdgpu.ids | tee ${O}_box_ragged.log
      rm -rf gpurun_out/prof_box
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_box -o box -- python $R/scripts/prof_box_ragged.py > /dev/null 2>&1)
      f=$(find gpurun_out/prof_box -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" ${O}_box_ragged_kernel_stats.csv && head -12 "$f" | cut -c1-160 ;;
    *) echo "unknown section $sec" ;;
  esac
done
