#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new template tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "matrix_core or template" 2>&1 | tail -3
echo "== property exploration, 200 examples: template, stencils (box ring), fast"; GS_HYPOTHESIS_EXAMPLES=200 timeout 1200 python -m pytest tests/test_property_shapes.py -m gpu -q -x --timeout 1100 -p no:cacheprovider -k "template or stencils or fast" 2>&1 | tail -4 | tee gpurun_out/property_explore2.log
