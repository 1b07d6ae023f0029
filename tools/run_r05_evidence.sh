#!/bin/bash
# round 5: evidence behind DESIGN.md section 4 "Round 5 -- GEMM" and profiles/r05_configs.md: (1) s_memtime phase timers of the persistent GEMM
# per shape and epilogue (bench library): share of K loop / drain / epilogue, epilogue cost in K-tile equivalents; (2) the kernel trace of
# BASELINE config 2 (576 x 512, batch 1) on the round-5 library.  usage: bash tools/run_r05_evidence.sh ; results gpurun_out/r05e/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05e; mkdir -p $out
make -C textflux_amd/csrc bench -j8 > $out/make_bench.log 2>&1
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so python tools/gemm_phase_timers.py --out $out/r05_gemm_phase_timers.json > $out/phase.log 2>&1
tail -9 $out/phase.log | cut -c1-400
rocprofv3 --kernel-trace --stats -d $out/tsl -o r05 --output-format csv -- python bench.py --no-cpu-baseline --no-pil-delta --no-attention-ab --no-graph --batch 1 --height 576 --width 512 --steps 1 --warmup 1 > $out/bench_sl512.log 2>&1
cp $out/tsl/*kernel_stats.csv $out/r05_sl512_b1_kernel_stats.csv 2>/dev/null; rm -rf $out/tsl
grep '^{"metric"' $out/bench_sl512.log > $out/r05_sl512_b1.json
head -12 $out/r05_sl512_b1_kernel_stats.csv | cut -c1-140
