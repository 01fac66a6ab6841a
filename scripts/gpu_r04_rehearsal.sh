#!/bin/bash
# round 4: the N = 2 code path of bench.py on a one-GPU box (gloo collectives, both ranks on GPU 0): weak scaling (2 x 128 frames)
# and --scaling strong (a 256-frame batch split 128 + 128); every frame of both ranks is checked against the golden file
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
export GS_BENCH_BACKEND=gloo GS_BENCH_DEVICE=0
for mode in weak strong; do
  F=$([ $mode = weak ] && echo 128 || echo 256)
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2964$([ $mode = weak ] && echo 1 || echo 2) \
    bench.py --gpus 2 --scaling $mode --frames $F --steps 5 --warmup 2 --no-cpu --no-other 2>gpurun_out/rehearsal_$mode.err | grep '^{' | tee gpurun_out/r04_bench_2rank_rehearsal_$mode.json | cut -c1-700
  tail -2 gpurun_out/rehearsal_$mode.err
done
