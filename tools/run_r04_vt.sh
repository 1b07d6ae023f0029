#!/bin/bash
# prices a transposed V tile in the attention kernel: W4_ABL=512 (one ds_read_b128 per V fragment instead of two transposing b64 reads; wrong results)
mkdir -p gpurun_out
for z in 1 0; do W4_ZERO=$z W4_ABL_LIST=512 timeout 300 python tools/attn_w4_ablate.py 31 32 2>&1 | grep lib; done | tee gpurun_out/r4_vt_ablation.log
