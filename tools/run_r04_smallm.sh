#!/bin/bash
# round 4: where the time goes at batch 1 (BASELINE config 2: 576x512; 1024x1024) -- kernel trace of eager pipeline calls, K-slicing levels 2 / 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04s; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
for v in "a 576 512" "b 576 512 --option gemm_splitk=1" "d 1024 1024" "e 1024 1024 --option gemm_splitk=1"; do
  set -- $v; tag=$1; h=$2; w=$3; shift 3
  rocprofv3 --kernel-trace --stats -d $out/t$tag -o r04 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-graph --batch 1 --height $h --width $w --steps 1 --warmup 1 "$@" > $out/bench_$tag.log 2>&1
  cp $out/t$tag/*kernel_stats.csv $out/r04_b1_${tag}_kernel_stats.csv 2>/dev/null
  grep '^{"metric"' $out/bench_$tag.log > $out/r04_b1_$tag.json
  rm -rf $out/t$tag
done
