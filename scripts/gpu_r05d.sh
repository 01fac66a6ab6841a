#!/bin/bash
# round 5, fourth visit: the whole GPU suite at the new state (k_lbp_tile by rule, libm ORB with device-side selection,
# GS_NO_STDLIB seam), the odd-stride A/B of the tile kernel, configs[4] through bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=5 2>&1 | tail -12 | tee gpurun_out/r05d_pytest_gpu.log
echo "== bench_lbp_tile quick (odd tile stride, default)"; timeout 600 python scripts/bench_lbp_tile.py quick 2>&1 | grep -v amdgpu.ids | grep -E "rule|cascade" | tee gpurun_out/r05d_lbp_tile_odd.log
echo "== bench_lbp_tile quick (even tile stride)"; UB_LIB=$R/build_variants/libgs_even_stride.so timeout 600 python scripts/bench_lbp_tile.py quick 2>&1 | grep -v amdgpu.ids | grep -E "rule|cascade" | tee gpurun_out/r05d_lbp_tile_even.log
echo "== configs[4] workload, 128 frames"
timeout 900 python bench.py --workload cfg4 --frames 128 --steps 1 --warmup 1 2>gpurun_out/r05d_cfg4.err | tee gpurun_out/r05d_cfg4_bench.json | cut -c1-1200
