#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_vae_kernels_gpu.py tests/test_production_gpu.py -x -q -k "conv or vae or VAE" 2>&1 | tail -6
timeout 300 python tools/vae_bench.py 2>&1 | tail -8
