#!/bin/bash
# One GPU-box visit: smoke, GPU parity suite, bench, rocprofv3 kernel stats.  Logs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=8 2>&1 | tail -60 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 2 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "== rocprofv3 stats"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; ls gpurun_out/prof 2>/dev/null | head; find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
