#!/bin/bash
# HBM traffic of every strip / pipeline kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate counter-only passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_*; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/scripts/pmc_probe.py > $R/gpurun_out/pmc_$c.log 2>&1
  cd $R; ls gpurun_out/pmc_$c | head -3
done
python scripts/pmc_summary.py gpurun_out 2>&1 | tee gpurun_out/pmc_summary.txt
