#!/bin/bash
# round 6 closing validation on one box: the whole GPU suite, the variant sweeps on the bench library, the default bench line, and the two rocprofv3
# passes over the bench whose summaries are committed (kernel-trace --stats of the eager run; the matrix-pipe PMC pass) -- untimed A/B calls and
# the live MFMA probe switched off so that the traces hold the product path only.  results under gpurun_out/r06f/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06f; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=10 ) > $out/gputests.log 2>&1
tail -18 $out/gputests.log
( time timeout 600 python -m pytest tools/variant_tests -q -m gpu ) > $out/variants.log 2>&1; tail -4 $out/variants.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/r06_bench_final.json
rocprofv3 --kernel-trace --stats -d $out/trace -o r06 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-peak-probe --no-graph --steps 1 --warmup 1 > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $out/bench_under_rocprof.log > $out/r06_bench_under_rocprof.json
cp $out/trace/*kernel_stats.csv $out/r06_bench_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o r06 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-peak-probe --no-graph --steps 1 --warmup 0 --denoise-steps 2 > $out/mfma.log 2>&1
python tools/pmc_bench_util.py $out/mfma/r06_counter_collection.csv $out/r06_mfma_util.json > $out/mfma_util.log 2>&1
rm -rf $out/mfma
python - <<'PY'
import json
for f in ("r06_bench_final.json", "r06_bench_under_rocprof.json"):
    d = json.loads(open("gpurun_out/r06f/" + f).read()); r = d["roofline"]
    print(f, d["value"], r["achieved"], r["frac"], r["dit_frac"], r["attention"]["achieved"], r.get("frac_of_capped"), (r.get("power_capped_peak") or {}).get("tflops"), (d.get("cpu_baseline") or {}).get("value"))
PY
head -8 $out/r06_bench_kernel_stats.csv | cut -c1-150
ls $out
