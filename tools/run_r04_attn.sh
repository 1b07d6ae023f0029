#!/bin/bash
# round 4: persistent attention -- tests, per-workgroup fixed cost, isolated timing, in-step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k attention 2>&1 | tail -25 > gpurun_out/r4a_tests.log
tail -4 gpurun_out/r4a_tests.log
timeout 300 python tools/attn_fixed_cost.py 2>&1 | tail -5
timeout 900 python tools/dit_ab.py attention_persistent=0,1 2>&1 | tail -2
