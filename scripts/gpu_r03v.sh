#!/bin/bash
# k_emit with a lane per 32-item half word (one trip per hit of the fullest half) instead of a visit per non-empty word
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
UB_ONLY=0 timeout 600 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fast_emit_halves.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ff -o st -- python $R/scripts/pmc_probe_fast.py > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_ff -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200 | tee -a gpurun_out/fast_emit_halves.log
echo "== gpu tests (emit users: FAST, ORB, match, LBP)"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fast or orb or keypoint or gsbatch or property or reference or match" 2>&1 | tail -3
