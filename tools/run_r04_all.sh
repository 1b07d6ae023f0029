#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4_gputests.log; tail -3 gpurun_out/r4_gputests.log
bash tools/run_r04_profiles.sh 2>&1 | tail -30
