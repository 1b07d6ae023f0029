#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/gemm4w_ab.py > gpurun_out/r4_gemm4w_ab.log 2>&1; cat gpurun_out/r4_gemm4w_ab.log | tail -8
timeout 900 python tools/dit_ab.py gemm_waves=8,4 2>&1 | tail -2
