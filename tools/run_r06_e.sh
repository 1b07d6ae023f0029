#!/bin/bash
# round 6: the fp8 fused-epilogue diagnosis test, then the round's profile collection (tools/run_r06_profiles.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k fp8_fused -s 2>&1 | tail -30 > gpurun_out/r06/fp8_fused_test.log; grep -n "differ\|passed\|failed\|Error" gpurun_out/r06/fp8_fused_test.log | cut -c1-400
bash tools/run_r06_profiles.sh > gpurun_out/r06/profiles.log 2>&1
tail -5 gpurun_out/r06/profiles.log; cat gpurun_out/r06/configs.log | tail -12
python - <<'PY'
import json
for f in ("r06_bench.json", "r06_bench_under_rocprof.json"):
    d = json.loads(open("gpurun_out/r06/" + f).read()); r = d["roofline"]
    print(f, d["value"], r["achieved"], r["frac"], r["dit_frac"], r["attention"]["achieved"], r.get("frac_of_capped"), (r.get("power_capped_peak") or {}).get("tflops"))
PY
head -8 gpurun_out/r06/r06_bench_kernel_stats.csv | cut -c1-160
