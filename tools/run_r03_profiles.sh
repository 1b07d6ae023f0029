#!/bin/bash
# Round-3 profile collection on one MI355X box (every --pmc pass is its own run, counters + kernel trace only).
# usage (from the repo root on the GPU box): bash tools/run_r03_profiles.sh ; results under gpurun_out/r03/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
# 1. kernel-trace summary of the default bench run (eager, so that every launch is in the trace) + its bench line
rocprofv3 --kernel-trace --stats -d $out/trace -o r03 --output-format csv -- python bench.py --no-cpu-baseline --no-graph --steps 1 --warmup 1 > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $out/bench_under_rocprof.log > $out/r03_bench_under_rocprof.json
cp $out/trace/*kernel_stats.csv $out/r03_bench_kernel_stats.csv 2>/dev/null
# 2. matrix-pipe busy over a 2-step bench
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/mfma -o r03 --output-format csv -- python bench.py --no-cpu-baseline --no-graph --steps 1 --warmup 0 --denoise-steps 2 > $out/mfma.log 2>&1
python tools/pmc_bench_util.py $out/mfma/r03_counter_collection.csv $out/r03_mfma_util.json > $out/mfma_util.log 2>&1
# 3. GEMM traffic at the three dominant shapes (separate passes)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/gemm_$n -o r03 --output-format csv -- python tools/bench_gemm_one.py > $out/gemm_$n.log 2>&1
done
# 4. attention: default kernel (30: 32x32x16 MFMA) vs the 16x16x32 kernel (40)
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/attn_$n -o r03 --output-format csv -- python tools/pmc_attn.py > $out/attn_$n.log 2>&1
done
python tools/r03_pmc_summary.py $out
# 5. power evidence: sustained loops with rocm-smi (kernel-level) and the register-only MFMA energy micro-benchmark
make -C textflux_amd/csrc bench -j8 > $out/make_bench.log 2>&1
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/power_profile.py --out $out/r03_power.json > $out/power.log 2>&1
python tools/mfma_power.py --secs 2.5 --out $out/r03_mfma_power.json > $out/mfma_power.log 2>&1
python tools/gemm_shapes_power.py --tag mfma16x16x32 --hipblaslt --out $out/r03_gemm_shapes.jsonl > $out/gemm_shapes.log 2>&1
# 5b. where a tile's time goes (s_memtime phase timers of the bench library), the q/k-norm epilogue shapes, VALU issue rates
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/gemm_phase_timers.py --out $out/r03_gemm_phase_timers.json > $out/phase.log 2>&1
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/qkn_ab.py > $out/r03_gemm_qkn_shapes.txt 2>&1
tools/ubench/valu_rate > $out/r03_valu_rate.jsonl 2>&1
# 6. cold start of the DiT (23.8 GB synthetic sharded checkpoint)
python tools/loader_bench.py --out $out/r03_loader.json > $out/loader.log 2>&1
# 7. BASELINE config 1 on the host cores (oracle, fp32, full 57-block model)
python bench.py --cpu-baseline-c1 > $out/r03_cpu_baseline_c1.json 2> $out/c1.err
ls $out
