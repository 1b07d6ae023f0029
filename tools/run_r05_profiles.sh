#!/bin/bash
# Round-5 profile collection on one MI355X box (every --pmc pass is its own run, counters + kernel trace only).
# usage (from the repo root on the GPU box): bash tools/run_r05_profiles.sh ; results under gpurun_out/r05/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05; mkdir -p $out
# 0. the default bench line (hipGraph replay) as the driver runs it
timeout 900 python bench.py > $out/bench.log 2>&1; grep '^{"metric"' $out/bench.log > $out/r05_bench.json
# 1. kernel-trace summary of the default bench run (eager, so that every launch is in the trace) + its bench line
rocprofv3 --kernel-trace --stats -d $out/trace -o r05 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-graph --steps 1 --warmup 1 > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $out/bench_under_rocprof.log > $out/r05_bench_under_rocprof.json
cp $out/trace/*kernel_stats.csv $out/r05_bench_kernel_stats.csv 2>/dev/null; rm -rf $out/trace
# 2. matrix-pipe busy over a 2-step bench
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o r05 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-graph --steps 1 --warmup 0 --denoise-steps 2 > $out/mfma.log 2>&1
python tools/pmc_bench_util.py $out/mfma/r05_counter_collection.csv $out/r05_mfma_util.json > $out/mfma_util.log 2>&1
# 3. GEMM traffic at the three dominant shapes (separate passes)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/gemm_$n -o r05 --output-format csv -- python tools/bench_gemm_one.py > $out/gemm_$n.log 2>&1
done
# 4. attention: the default kernel with and without the score bound
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/attn_$n -o r05 --output-format csv -- python tools/pmc_attn.py > $out/attn_$n.log 2>&1
done
python tools/r05_pmc_summary.py $out > $out/pmc_summary.log 2>&1
# 5. power evidence + the six GEMM shapes with their epilogues next to hipBLASLt
make -C textflux_amd/csrc bench -j8 > $out/make_bench.log 2>&1
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/power_profile.py --out $out/r05_power.json > $out/power.log 2>&1
python tools/gemm_shapes_power.py --tag r05 --hipblaslt --out $out/r05_gemm_shapes.jsonl > $out/gemm_shapes.log 2>&1
# 6. other BASELINE geometries / batch sizes / precisions
bash tools/run_configs.sh > $out/configs.log 2>&1; cp gpurun_out/configs.jsonl $out/r05_configs.jsonl
# 7. BASELINE config 1 on the host cores (oracle, fp32, full 57-block model)
python bench.py --cpu-baseline-c1 > $out/r05_cpu_baseline_c1.json 2> $out/c1.err
rm -rf $out/mfma $out/gemm_* $out/attn_*
ls $out
