#!/bin/bash
# A/B of the sobel instruction-count changes (build_variants/libgs_old.so = before) + GPU suite + gsbatch timing
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
echo "== A/B 3840x2160 x64"; AB_TAGS=base,old,nosat UB_OPS=sobel,bs,fused,copy,erode timeout 300 python scripts/ubench_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_sobel_4k.log
echo "== A/B 4096x4096 x64"; UB_W=4096 UB_H=4096 AB_TAGS=base,old,nosat UB_OPS=sobel,bs,copy timeout 300 python scripts/ubench_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_sobel_4096.log
echo "== gsbatch, 64 x 3840x2160 PGM files in /dev/shm"
python - <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
from oracle.pyoracle import Oracle
os.makedirs("/dev/shm/gb_in", exist_ok=True); os.makedirs("/dev/shm/gb_out", exist_ok=True)
for k in range(64):
    a = Oracle.synth(3840, 2160, 1000 + k)
    with open("/dev/shm/gb_in/f%03d.pgm" % k, "wb") as f:
        f.write(b"P5\n3840 2160\n255\n"); f.write(a.tobytes())
P
make -s -C grayskull_amd/csrc tool
for i in 1 2; do
  ( time ./grayskull_amd/gsbatch -v -o /dev/shm/gb_out blur 2 : sobel : threshold otsu : morph dilate 2 -- /dev/shm/gb_in/*.pgm ) 2>&1 | tail -6
done | tee gpurun_out/gsbatch_64x4k.log
