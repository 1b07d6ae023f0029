#!/bin/bash
# round 4 closing validation: the whole GPU suite + the default bench line on the final library
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r4_final_gputests.log
tail -3 gpurun_out/r4_final_gputests.log
timeout 900 python bench.py > gpurun_out/r4_final_bench.log 2>&1; grep '^{"metric"' gpurun_out/r4_final_bench.log > gpurun_out/r4_final_bench.json
python -c "import sys,json; d=json.loads(open('gpurun_out/r4_final_bench.json').read()); print(d['value'], d['roofline']['frac'], d['roofline']['dit_frac'], d['roofline']['attention'], d['pil_output_delta_ms_per_call'])"
