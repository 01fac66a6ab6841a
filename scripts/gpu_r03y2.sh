#!/bin/bash
# timing experiments on k_match_template_mfma: builds with one phase removed (GS_TM_VARIANT 1..5; wrong results on purpose)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for l in "" build_variants/libgs_tm1.so build_variants/libgs_tm2.so build_variants/libgs_tm3.so build_variants/libgs_tm4.so build_variants/libgs_tm5.so; do
  UB_LIB=${l:+$R/$l} timeout 300 python scripts/ubench_tmatch_variants.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/tmatch_variants.log
