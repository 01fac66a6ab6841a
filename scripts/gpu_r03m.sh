#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/ubench_fast_nms.py 2>&1 | grep -v amdgpu.ids | grep "x32 cap 2000\|x8 cap" | tee gpurun_out/fast_nms2.log
echo "== gpu tests (all but lbp)"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "not lbp and not cascade and not config4 and not dist" 2>&1 | tail -3
echo "== kernel times"; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast -o stats -- python $R/scripts/pmc_probe_features.py > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_fast -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "fast|emit|scan|Name|fill" "$f" | cut -c1-40,100-190 | tee gpurun_out/fast_kernel_stats2.txt
