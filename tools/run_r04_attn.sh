#!/bin/bash
# round 4: attention -- tests (incl. determinism soak), isolated timing, in-step timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -k "attention" 2>&1 | tail -3
timeout 600 python tools/attn_w4_soak.py 2>&1 | tail -6
W4_NO_ABL=1 timeout 300 python tools/attn_w4_ablate.py 30 34 2>&1 | grep lib
timeout 900 python tools/dit_ab.py attention_use_bound=0,1 2>&1 | tail -2
