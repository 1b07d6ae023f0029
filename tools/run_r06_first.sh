#!/bin/bash
# round 6, first GPU call: the trimmed suite with durations, the variant sweeps on the bench library, a default bench line (live power-cap
# probe + live C1), the honest GEMM-shape table and the q-side A/B.  results under gpurun_out/r06a/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06a; mkdir -p $out
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=60 ) > $out/gputests.log 2>&1
tail -75 $out/gputests.log | head -70
( time timeout 900 python -m pytest tools/variant_tests -x -q -m gpu ) > $out/variants.log 2>&1; tail -5 $out/variants.log
( time timeout 900 python bench.py --steps 3 --warmup 2 ) > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/r06_bench_first.json; tail -3 $out/bench.log | cut -c1-600
timeout 600 python tools/gemm_shapes_power.py --tag r06 --hipblaslt --out $out/r06_gemm_shapes.jsonl > $out/shapes.log 2>&1; tail -2 $out/shapes.log
timeout 300 python tools/qkn_ab6.py $out/r06_qkn_ab.json > $out/qkn_ab.log 2>&1; tail -12 $out/qkn_ab.log
ls $out
