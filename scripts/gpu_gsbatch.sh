#!/bin/bash
# gsbatch on the GPU box: parity test + timing over 64 4K PGM files in /dev/shm
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
true
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
for sl in 33554432 67108864 134217728 268435456 1073741824; do export GSBATCH_SLICE_BYTES=$sl; echo "slice $sl"
  ( time ./grayskull_amd/gsbatch -v -o /dev/shm/gb_out blur 2 : sobel : threshold otsu : morph dilate 2 -- /dev/shm/gb_in/*.pgm ) 2>&1 | tail -6
done | tee gpurun_out/gsbatch_64x4k.log
ls /dev/shm/gb_out | wc -l
