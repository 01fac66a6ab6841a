#!/bin/bash
# One GPU-box visit: smoke, GPU parity suite, bench, rocprofv3 kernel stats + PMC traffic.  Logs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/pmc_*
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=5 2>&1 | tail -12 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -3 gpurun_out/bench.err
echo "== rocprofv3 kernel stats (same bench command)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --no-cpu --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
echo "== rocprofv3 PMC passes (separate runs, counters only)"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/scripts/pmc_probe.py > $R/gpurun_out/pmc_$c.log 2>&1
  cd $R; ls gpurun_out/pmc_$c | head -5
done
python scripts/pmc_summary.py gpurun_out 2>&1 | tee gpurun_out/pmc_summary.txt
