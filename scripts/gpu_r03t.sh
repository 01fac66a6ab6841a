#!/bin/bash
# gs_fast: k_fast_fused (score + NMS + mask words in one walk) (key 7 = 6) vs the two-pass form (default)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== FAST/ORB gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "fast or orb or keypoint or gsbatch or property or reference or match" 2>&1 | tail -3
UB_ONLY=6,0 timeout 600 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fast_fused.log
for m in 2 3 6 8 16; do echo "== m = $m (key 0)"; UB_M=$m UB_ONLY=6 UB_SYNTH_ONLY=1 timeout 300 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | head -2; done | tee gpurun_out/fast_fused_m.log
echo "== 4K x 8 and 1080p x 8"
UB_W=3840 UB_H=2160 UB_F=8 UB_ONLY=6,0 UB_SYNTH_ONLY=1 timeout 300 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | head -3 | tee -a gpurun_out/fast_fused.log
UB_W=1920 UB_H=1080 UB_F=8 UB_ONLY=6,0 UB_SYNTH_ONLY=1 timeout 300 python scripts/ubench_fast.py 2>&1 | grep -v amdgpu.ids | head -3 | tee -a gpurun_out/fast_fused.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ff -o st -- python $R/scripts/pmc_probe_fast.py > /dev/null 2>&1; cd $R
f=$(find gpurun_out/prof_ff -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-160 | tee gpurun_out/fast_fused_stats.txt
