#!/bin/bash
# profiles/fused_isa_mix.json from hipcc's assembly of the fused pipeline kernel (no GPU needed); run after every change of
# k_fused.h / gs_fused.cpp / k_strip.h (tests/test_bench_contract.py checks the stamp)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/gs_asm && cd /tmp/gs_asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=iterative-ilp -x hip -c $R/grayskull_amd/csrc/gs_fused.cpp -save-temps -o gs_fused.o 2>/dev/null
cd $R
python scripts/isa_count.py /tmp/gs_asm/gs_fused-hip-amdgcn-amd-amdhsa-gfx950.s k_blur_sobel_hist16ILi2ELb1ELi0E 6 profiles/fused_isa_mix.json | tail -3
python scripts/stamp.py profiles/fused_isa_mix.json grayskull_amd/csrc/k_fused.h grayskull_amd/csrc/gs_fused.cpp grayskull_amd/csrc/k_strip.h
