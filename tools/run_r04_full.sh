#!/bin/bash
# round 4 mid-round validation: the whole GPU suite, the full-depth 30-step trajectory (engine side), the default bench line, batch-1 rows
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r4_gputests.log
tail -3 gpurun_out/r4_gputests.log
timeout 900 python tools/fulldepth_trajectory.py --engine > gpurun_out/r4_fulldepth.log 2>&1; tail -2 gpurun_out/r4_fulldepth.log
timeout 900 python bench.py > gpurun_out/r4_bench.log 2>&1; grep '^{"metric"' gpurun_out/r4_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['dit_frac'], d['roofline']['attention'], d['pil_output_delta_ms_per_call'])"
for g in "576 512" "1024 1024"; do set -- $g; timeout 600 python bench.py --no-cpu-baseline --no-pil-delta --batch 1 --height $1 --width $2 2>&1 | grep '^{"metric"' > gpurun_out/r4_b1_$1.json; python -c "import sys,json; d=json.loads(open('gpurun_out/r4_b1_$1.json').read()); print('$1', d['value'], d['dit_algorithmic_tflops_per_gpu'], d['roofline']['achieved'], d['roofline']['attention']['achieved'])"; done
