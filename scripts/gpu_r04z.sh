#!/bin/bash
# round 4, full visit: smoke, GPU suite, bench (plain / under RCCL at world 1 / configs[4] / the 4096-frame batch on one GPU),
# rocprofv3 kernel stats of the full bench command, PMC traffic, the north-star launch alone, ragged shapes, box offsets.
# Logs -> gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/prof gpurun_out/pmc_* gpurun_out/sqfeat_*; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=6 2>&1 | tail -14 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json | cut -c1-500
tail -3 gpurun_out/bench.err
echo "== bench under torch.distributed.run, 1 rank, GS_BENCH_FORCE_DIST=1 (nccl == RCCL)"
GS_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --no-other --no-cpu --steps 20 2>gpurun_out/bench_rccl.err | grep '^{' | tee gpurun_out/bench_rccl_world1.json | cut -c1-300
echo "== configs[4] workload, 128 frames, golden + live oracle check of frame 0"
timeout 900 python bench.py --workload cfg4 --frames 128 --steps 1 --warmup 1 --verify-live 2>gpurun_out/cfg4.err | tee gpurun_out/cfg4_bench.json | cut -c1-900
echo "== the whole 4096-frame batch on this one GPU (strong-scaling denominator)"
timeout 900 python bench.py --scaling strong --frames 4096 --steps 10 --warmup 2 --no-cpu --no-other 2>gpurun_out/strong.err | tee gpurun_out/bench_strong4096.json | cut -c1-400
echo "== rocprofv3 kernel stats (the full bench command)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o stats -- python $R/bench.py --no-cpu --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cd $R; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats.csv && head -30 "$f" | cut -c1-150
echo "== PMC: HBM traffic per launch"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/scripts/pmc_probe.py > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R; python scripts/pmc_summary.py gpurun_out 2>&1 | tee gpurun_out/pmc_summary.txt | tail -40
echo "== PMC: feature kernels (VALU instructions of the gs_fast passes)"
PMC_PROBE=scripts/pmc_probe_features.py PMC_FILTER=k_fast,k_emit PMC_TAG=sqfeat \
  PMC_SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS" bash scripts/pmc_fused.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pmc_features.txt
python scripts/pmc_fast_json.py > gpurun_out/pmc_fast_json.log 2>&1; tail -2 gpurun_out/pmc_fast_json.log | cut -c1-300
echo "== north-star launch alone"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sobel4096 -o sobel4096 -- python $R/scripts/prof_sobel4096.py 2>&1 | grep "gs_sobel 64" | tee $R/gpurun_out/sobel4096.log
cd $R; f=$(find gpurun_out/prof_sobel4096 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/sobel4096_kernel_stats.csv && head -3 "$f" | cut -c1-200
echo "== ragged shapes"; RG_CHECK=0 timeout 500 python scripts/ubench_ragged.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ragged.log | tail -40
echo "== box offsets"; timeout 300 python scripts/ubench_box_offsets.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/box_offsets.log
echo "== next rows"; timeout 300 python scripts/ubench_next_rows.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/next_rows.log | tail -12
echo "== gs_match_template: matrix cores vs dot-product kernels"; timeout 300 python scripts/ubench_tmatch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tmatch.log | tail -12
echo "== gs_fast, 32 x 720p: score pass and whole call, sparse vs strip NMS"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fast.log
import sys, os
sys.path.insert(0, os.getcwd())
import torch, grayskull_amd as gs
g = gs.lib(); g.use_torch_stream()
def timeit(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nf, h, w = 32, 720, 1280
f = torch.empty((nf, h, w), dtype=torch.uint8, device="cuda"); g.synth_batch(f, 4)
flat = torch.full((nf, h, w), 100, dtype=torch.uint8, device="cuda")
rnd = torch.randint(0, 256, (nf, h, w), dtype=torch.uint8, device="cuda")
sm = torch.zeros((nf, h, w), dtype=torch.uint8, device="cuda")
kp = torch.zeros((nf, 2000, 12), dtype=torch.int32, device="cuda"); cn = torch.zeros(nf, dtype=torch.int32, device="cuda")
for key19 in (0, 2, 0, 2):
    g.tune(19, key19)
    print("NMS %s: score us: block noise %.1f flat %.1f random %.1f | gs_fast us: block noise %.1f flat %.1f random %.1f" % (
        "sparse" if key19 == 0 else "strips", timeit(lambda: g.probe_fast_score(sm, f, 20)), timeit(lambda: g.probe_fast_score(sm, flat, 20)),
        timeit(lambda: g.probe_fast_score(sm, rnd, 20)), timeit(lambda: g.fast_batch(f, sm, kp, cn, 2000, 20)),
        timeit(lambda: g.fast_batch(flat, sm, kp, cn, 2000, 20)), timeit(lambda: g.fast_batch(rnd, sm, kp, cn, 2000, 20))), flush=True)
g.tune(19, 0)
PY
