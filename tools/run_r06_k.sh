#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06k
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_bench.so timeout 600 python tools/power_profile.py --out gpurun_out/r06k/r06_power.json --secs 2.5 2>&1 | grep -v "^$" | cut -c1-220 | tail -20
