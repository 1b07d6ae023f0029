#!/bin/bash
# Re-measures the rows of profiles/r0N_configs.md (one bench.py line each) into gpurun_out/configs.jsonl
out=gpurun_out/configs.jsonl; : > $out
run() { timeout 600 python bench.py --no-cpu-baseline --no-attention-ab --no-peak-probe "$@" 2>&1 | grep '^{"metric"' | tail -1 >> $out; }
run --batch 1
run --batch 1 --height 576 --width 512
run --batch 1 --height 1184 --width 1024
run --batch 1 --height 2048 --width 1024
run --batch 8 --height 2048 --width 1024
run --batch 4 --height 1184 --width 1024 --sampler amo
run --fp8 --batch 1
run --fp8
run --fp8 --denoise-steps 50
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:70], "| s/img", round(d["sec_per_img_per_gpu"], 3), "| img/s", round(d["value"], 3), "| DiT TF", round(d["dit_algorithmic_tflops_per_gpu"] or 0), "| gemm", round(d["roofline"]["achieved"]), "| attn", round(d["roofline"]["attention"]["achieved"]))
PY
