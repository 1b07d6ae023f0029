#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06j; mkdir -p $out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "probe or per_batch" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats -d $out/trace -o r06 --output-format csv -- python bench.py --fp8 --no-cpu-baseline --no-pil-delta --no-attention-ab --no-peak-probe --no-graph --steps 1 --warmup 1 > $out/fp8.log 2>&1
cp $out/trace/*kernel_stats.csv $out/r06_fp8_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
head -16 $out/r06_fp8_kernel_stats.csv | python -c "
import sys,csv
for r in csv.DictReader(sys.stdin): print(r['Name'][:78].ljust(78), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])"
