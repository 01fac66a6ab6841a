#!/bin/bash
# round 3, visit A: GPU suite (incl. the RCCL world-1 tests), bench, LBP prefilter A/B, ordering-event A/B,
# rocprofv3 kernel stats of the full bench command.  Logs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/pmc_*
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== LBP prefilter"; timeout 600 python scripts/bench_lbp_pre.py -1,1,2,3,4,102 2>&1 | tee gpurun_out/lbp_pre.log | tail -40
echo "== LBP prefilter, global-load variant"; UB_LIB=$R/build_variants/libgs_dense_global.so timeout 300 python scripts/bench_lbp_pre.py 2,3 2>&1 | tee gpurun_out/lbp_pre_global.log | tail -14
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider --durations=8 2>&1 | tail -16 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-600
tail -3 gpurun_out/bench.err
echo "== ordering events: default (release) vs DisableSystemFence"
for v in base orderfence base orderfence; do
  if [ $v = base ]; then unset GS_BENCH_LIB; else export GS_BENCH_LIB=$R/build_variants/libgs_$v.so; fi
  timeout 300 python bench.py --no-cpu --no-other --no-verify --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" | tee -a gpurun_out/order_events.log
done
unset GS_BENCH_LIB
echo "== rocprofv3 kernel stats (full bench command)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --no-cpu --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-160
