#!/bin/bash
# Round-6 profile collection on one MI355X box (every --pmc pass is its own run, counters + kernel trace only).
# usage (from the repo root on the GPU box): bash tools/run_r06_profiles.sh ; results under gpurun_out/r06/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06; mkdir -p $out
# 0. the default bench line (hipGraph replay; live power-capped peak + live C1 cpu_baseline) as the driver runs it
timeout 900 python bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/r06_bench.json
# 1. kernel-trace summary of the default bench run (eager, so that every launch is in the trace) + its bench line
rocprofv3 --kernel-trace --stats -d $out/trace -o r06 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-peak-probe --no-graph --steps 1 --warmup 1 > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $out/bench_under_rocprof.log > $out/r06_bench_under_rocprof.json
cp $out/trace/*kernel_stats.csv $out/r06_bench_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
# 2. matrix-pipe busy over a 2-step bench
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o r06 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-peak-probe --no-graph --steps 1 --warmup 0 --denoise-steps 2 > $out/mfma.log 2>&1
python tools/pmc_bench_util.py $out/mfma/r06_counter_collection.csv $out/r06_mfma_util.json > $out/mfma_util.log 2>&1
# 3. GEMM traffic at the three dominant shapes (separate passes)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/gemm_$n -o r06 --output-format csv -- python tools/bench_gemm_one.py > $out/gemm_$n.log 2>&1
done
# 4. attention: the default kernel with and without the score bound (+ a finer wait split, round 6)
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/attn_$n -o r06 --output-format csv -- python tools/pmc_attn.py > $out/attn_$n.log 2>&1
done
python tools/r06_pmc_summary.py $out > $out/pmc_summary.log 2>&1
cp $out/attn_SQ_WAIT_INST_LDS/r06_counter_collection.csv $out/r06_attention_wait_split_raw.csv 2>/dev/null
# 5. the six GEMM shapes with the epilogue each carries in the model next to hipBLASLt (bias only / + the passes it then needs)
python tools/gemm_shapes_power.py --tag r06 --hipblaslt --out $out/r06_gemm_shapes.jsonl > $out/gemm_shapes.log 2>&1
# 6. other BASELINE geometries / batch sizes / precisions
bash tools/run_configs.sh > $out/configs.log 2>&1; cp gpurun_out/configs.jsonl $out/r06_configs.jsonl
rm -rf $out/mfma $out/gemm_FETCH_SIZE $out/gemm_WRITE_SIZE $out/gemm_SQ_VALU_MFMA_BUSY_CYCLES $out/attn_SQ_*
ls $out
