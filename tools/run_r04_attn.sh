#!/bin/bash
# round 4: attention modes 30 .. 33 (attn_w4_kernel<0..3>) -- tests, timing ablations on zero (cycle-bound) and random (power-bound) data
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k attention 2>&1 | tail -25 > gpurun_out/r4a_tests.log
W4_ZERO=1 timeout 600 python tools/attn_w4_ablate.py 30 33 32 > gpurun_out/r4a_ablate_zero.log 2>&1
timeout 600 python tools/attn_w4_ablate.py 30 33 > gpurun_out/r4a_ablate.log 2>&1
tail -5 gpurun_out/r4a_tests.log; grep lib gpurun_out/r4a_ablate_zero.log; grep lib gpurun_out/r4a_ablate.log
