#!/bin/bash
# round 4: attention modes 30 .. 34 (attn_w4_kernel<0..4>) -- tests, isolated timing on zero / random data, in-step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k attention 2>&1 | tail -25 > gpurun_out/r4a_tests.log
W4_NO_ABL=1 W4_ZERO=1 timeout 300 python tools/attn_w4_ablate.py 30 33 34 > gpurun_out/r4a_modes_zero.log 2>&1
W4_NO_ABL=1 timeout 300 python tools/attn_w4_ablate.py 30 33 34 > gpurun_out/r4a_modes.log 2>&1
timeout 900 python tools/dit_ab.py attention_use_bound=0,1 > gpurun_out/r4a_dit.log 2>&1
tail -5 gpurun_out/r4a_tests.log; grep lib gpurun_out/r4a_modes_zero.log gpurun_out/r4a_modes.log; tail -3 gpurun_out/r4a_dit.log
