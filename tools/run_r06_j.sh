#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06j; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o r06 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-peak-probe --no-graph --steps 1 --warmup 1 --batch 1 --height 576 --width 512 > $out/sl512.log 2>&1
cp $out/trace/*kernel_stats.csv $out/r06_sl512_b1_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
head -14 $out/r06_sl512_b1_kernel_stats.csv | python -c "
import sys,csv
for r in csv.DictReader(sys.stdin): print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])"
