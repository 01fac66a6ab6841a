#!/bin/bash
# round-2 GPU visit: smoke, parity suite, bench (+ rocprofv3 kernel stats of the same command), 2-rank rehearsal
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider --durations=6 2>&1 | tail -14 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-600
tail -3 gpurun_out/bench.err
echo "== rocprofv3 kernel stats (same bench command, no extras)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --no-cpu --no-verify --no-other > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160
echo "== --gpus 2 on a 1-GPU box must refuse"; python bench.py --gpus 2 --steps 2 2>&1 | tail -1
echo "== 2-rank rehearsal (gloo, both ranks share the one GPU: the number means nothing)"
GS_BENCH_BACKEND=gloo GS_BENCH_DEVICE=0 timeout 600 python bench.py --gpus 2 --frames 64 --steps 5 --no-cpu 2>gpurun_out/rehearsal.err | tee gpurun_out/bench_2rank_rehearsal.json | cut -c1-400
tail -2 gpurun_out/rehearsal.err
