#!/bin/bash
# round 4, visit f: full GPU suite, default bench line, the 4096-frame batch on ONE GPU (strong-scaling denominator, all
# 4096 frames against the golden checksums), configs[4] with the extended golden set.  Logs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r04f_smoke.log
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 2>&1 | tail -16 | tee gpurun_out/r04f_pytest_gpu.log
echo "== bench (default)"; timeout 900 python bench.py 2>gpurun_out/r04f_bench.err | tee gpurun_out/r04f_bench.json | cut -c1-600
tail -3 gpurun_out/r04f_bench.err
echo "== bench --scaling strong --frames 4096 (N = 1 holds the whole batch)"
timeout 900 python bench.py --scaling strong --frames 4096 --steps 10 --warmup 2 --no-cpu --no-other 2>gpurun_out/r04f_strong.err | tee gpurun_out/r04f_bench_strong4096.json | cut -c1-400
tail -3 gpurun_out/r04f_strong.err
echo "== configs[4] workload, 128 frames"
timeout 900 python bench.py --workload cfg4 --frames 128 --steps 1 --warmup 1 2>gpurun_out/r04f_cfg4.err | tee gpurun_out/r04f_cfg4_bench.json | cut -c1-900
tail -3 gpurun_out/r04f_cfg4.err
