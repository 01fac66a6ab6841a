#!/bin/bash
# round-2 GPU visit: parity suite, A/B of the fused-kernel variants, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
echo "== A/B 3840x2160 x64 (base = this tree, r01 = round-1 kernels)"
[ -f build_variants/libgs_base.so ] && mv build_variants/libgs_base.so build_variants/libgs_r01.so
AB_TAGS=base,r01 UB_OPS=fused,bs,sobel,blur2 timeout 300 python scripts/ubench_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_fused.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-1200
tail -3 gpurun_out/bench.err
