#!/bin/bash
# round 4: where the time goes at batch 1 (BASELINE config 2: 576x512, and 1024x1024) -- kernel trace of one eager pipeline call each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04s; mkdir -p $out
for g in "576 512" "1024 1024"; do
  set -- $g
  rocprofv3 --kernel-trace --stats -d $out/t$1 -o r04 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-graph --batch 1 --height $1 --width $2 --steps 1 --warmup 1 > $out/bench_$1.log 2>&1
  cp $out/t$1/*kernel_stats.csv $out/r04_b1_$1_kernel_stats.csv 2>/dev/null
  grep '^{"metric"' $out/bench_$1.log > $out/r04_b1_$1.json
  rm -rf $out/t$1
done
head -14 $out/r04_b1_576_kernel_stats.csv | cut -c1-200
