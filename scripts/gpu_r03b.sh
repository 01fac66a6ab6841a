#!/bin/bash
# round 3, visit B: where the cascade's time goes (truncated cascades) + prefetch depth of k_lbp_dense
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== stages"; timeout 900 python scripts/bench_lbp_stages.py 1,2,3,4,5,6,8,10,14,20 -1,2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_stages.log
for v in pd1 base pd3 pd2_global; do
  echo "== dense prefetch variant $v"
  if [ $v = base ]; then unset UB_LIB; else export UB_LIB=$R/build_variants/libgs_$v.so; fi
  timeout 300 python scripts/bench_lbp_stages.py 2,4 2,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_stages_$v.log
done
unset UB_LIB
echo "== prefilter"; timeout 600 python scripts/bench_lbp_pre.py -1,2,3,4,5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lbp_pre.log | head -24
