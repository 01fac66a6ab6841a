#!/bin/bash
# round 3, full visit: smoke, GPU suite, bench (plain / under RCCL at world 1 / configs[4]), rocprofv3 kernel stats of the
# full bench command, PMC traffic + feature counters + LBP counters.  Logs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/pmc_* gpurun_out/sqfeat_* gpurun_out/sqbox_*; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=6 2>&1 | tail -14 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-500
tail -3 gpurun_out/bench.err
echo "== bench under torch.distributed.run, 1 rank, GS_BENCH_FORCE_DIST=1 (nccl == RCCL)"
GS_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --no-other --no-cpu --steps 20 2>gpurun_out/bench_rccl.err | grep '^{' | tee gpurun_out/bench_rccl_world1.json | cut -c1-300
echo "== configs[4] workload, 64 frames, golden + live oracle check of frame 0"
timeout 900 python bench.py --workload cfg4 --frames 64 --steps 2 --warmup 1 --verify-live 2>gpurun_out/cfg4.err | tee gpurun_out/cfg4_bench.json | cut -c1-900
echo "== A/B build with compiler-visible LDS atomics in the fused kernel (-DGS_PLAIN_LDS_ATOMICS): same bytes (golden parity), time"
[ -f build_variants/libgs_plain_lds.so ] && GS_BENCH_LIB=$R/build_variants/libgs_plain_lds.so timeout 600 python bench.py --no-cpu --no-other --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain_lds', d['value'], d['ms_per_step'], d['parity'][:90])" | tee gpurun_out/plain_lds_ab.log
echo "== --gpus 2 on a 1-GPU box must refuse"; python bench.py --gpus 2 --steps 2 2>&1 | tail -1
echo "== rocprofv3 kernel stats (the full bench command)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --no-cpu --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-150
echo "== PMC: HBM traffic per launch"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/scripts/pmc_probe.py > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R; python scripts/pmc_summary.py gpurun_out 2>&1 | tee gpurun_out/pmc_summary.txt | tail -40
echo "== PMC: feature kernels"
PMC_PROBE=scripts/pmc_probe_features.py PMC_FILTER=k_fast,k_hist_partial,k_lbp,k_emit,k_chunk PMC_TAG=sqfeat \
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_features.txt
python scripts/pmc_fast_json.py > gpurun_out/pmc_fast_json.log 2>&1; tail -2 gpurun_out/pmc_fast_json.log | cut -c1-300
echo "== PMC: sliding box (r = 8) and integral kernels, 64 x 4K"
PMC_PROBE=scripts/pmc_probe_box.py PMC_FILTER=k_box16,k_integral PMC_TAG=sqbox bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_box.txt
echo "== PMC: LBP cascade, 8 x 4K noise"
LBP_PRE=0 PMC_SETS="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum|TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" bash scripts/pmc_lbp.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_lbp_quad.txt
echo "== next rows"; timeout 300 python scripts/ubench_next_rows.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/next_rows.log | tail -12
echo "== gs_match_template: matrix cores vs dot-product kernels"; timeout 600 python scripts/ubench_tmatch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tmatch_mfma.log | tail -12
