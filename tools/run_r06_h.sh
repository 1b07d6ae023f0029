#!/bin/bash
# the bench as the driver ran it in round 5 (20 timed calls, 5 warm-up calls): wall time of the whole command incl. the live C1 subprocess
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06h
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06h/bench_driverlike.log 2>&1
grep '^{"metric"' gpurun_out/r06h/bench_driverlike.log > gpurun_out/r06h/r06_bench_driverlike.json
tail -4 gpurun_out/r06h/bench_driverlike.log | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06h/r06_bench_driverlike.json").read()); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["dit_frac"], r["attention"]["achieved"], r["frac_of_capped"], d["cpu_baseline"]["value"], d["cpu_baseline"]["c1"]["wall_s_incl_weights_and_warmup"])
PY
