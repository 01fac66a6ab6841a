#!/bin/bash
# k_box16r with amdgpu_waves_per_eu(r <= 12 ? 4 : 2) (a few scratch spills, 4 waves per SIMD) vs the plain build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== plain build"; UB_BIG_ONLY=1 timeout 600 python scripts/ubench_box_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/box_ring_plain.log
echo "== waves_per_eu build"; UB_BIG_ONLY=1 UB_LIB=$R/build_variants/libgs_boxattr.so timeout 600 python scripts/ubench_box_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/box_ring_attr.log
