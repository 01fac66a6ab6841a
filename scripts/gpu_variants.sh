#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for v in base ntld ntst ntboth; do
  echo "=== variant $v"
  if [ $v = base ]; then unset UB_LIB; else export UB_LIB=$PWD/build_variants/libgs_$v.so; fi
  UB_PF=1 UB_T=${UB_T:-0,32} UB_G=${UB_G:-1} python scripts/ubench.py 2>&1 | tee gpurun_out/ubench_$v.log | grep -E "^(copy|erode|sobel|blur2) |threshold|torch"
done
