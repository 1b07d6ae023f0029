#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06g
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_vae_kernels_gpu.py tests/test_production_gpu.py tests/test_pipeline_gpu.py tests/test_e2e_gpu.py -q -m gpu -x 2>&1 | tail -15 | cut -c1-300
timeout 300 python tools/vae_bench.py 2>&1 | tail -6 | cut -c1-300
