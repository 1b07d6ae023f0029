#!/bin/bash
# round 4: attention -- tests (incl. determinism soak), per-item phase timers, isolated timing, in-step timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -k "attention" 2>&1 | tail -3
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so timeout 300 python tools/attn_item_timers.py 2>&1 | grep data | head -4
W4_NO_ABL=1 W4_ZERO=1 timeout 300 python tools/attn_w4_ablate.py 34 2>&1 | grep lib
W4_NO_ABL=1 timeout 300 python tools/attn_w4_ablate.py 34 2>&1 | grep lib
timeout 900 python tools/dit_ab.py attention_persistent=0,1 2>&1 | tail -2
