#!/bin/bash
# round 5 closing validation on one box: the whole GPU suite, the default bench line, and the two rocprofv3 passes over the bench whose
# summaries are committed (kernel-trace --stats of the eager run; the matrix-pipe PMC pass) -- with the untimed attention A/B calls
# switched off so that the traces hold the product path only.  usage: bash tools/run_r05_final.sh ; results under gpurun_out/r05f/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05f; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > $out/gputests.log
tail -3 $out/gputests.log
timeout 900 python bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/r05_bench_final.json
rocprofv3 --kernel-trace --stats -d $out/trace -o r05 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-graph --steps 1 --warmup 1 > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $out/bench_under_rocprof.log > $out/r05_bench_under_rocprof.json
cp $out/trace/*kernel_stats.csv $out/r05_bench_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o r05 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-graph --steps 1 --warmup 0 --denoise-steps 2 > $out/mfma.log 2>&1
python tools/pmc_bench_util.py $out/mfma/r05_counter_collection.csv $out/r05_mfma_util.json > $out/mfma_util.log 2>&1
rm -rf $out/mfma
python -c "import json; d=json.loads(open('$out/r05_bench_final.json').read()); print(d['value'], d['roofline']['frac'], d['roofline']['dit_frac'], d['roofline']['attention']['frac'], d['roofline']['attention']['call_s_bound_ignored'], d['cpu_baseline']['sample'][:120])"
ls $out
