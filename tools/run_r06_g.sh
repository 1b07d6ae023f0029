#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06g
timeout 900 python -m pytest tests/test_vae_kernels_gpu.py tests/test_production_gpu.py -q -m gpu -k "blend or tiling or tiled" -s 2>&1 | tail -25 | cut -c1-300
