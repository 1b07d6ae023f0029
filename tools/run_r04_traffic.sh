#!/bin/bash
# re-collects the GEMM traffic counters after the output stores became non-temporal, and the attention kernel's FETCH_SIZE / WRITE_SIZE (separate
# --pmc passes, counters + kernel trace only); usage (GPU box, repo root): bash tools/run_r04_traffic.sh ; results under gpurun_out/r04b/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04b; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace -d $out/gemm_$n -o r04 --output-format csv -- python tools/bench_gemm_one.py > $out/gemm_$n.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/attnio_$c -o r04 --output-format csv -- python tools/pmc_attn.py > $out/attnio_$c.log 2>&1
done
python tools/r04_pmc_summary.py $out > $out/pmc_summary.log 2>&1
python - <<'PY'
import csv, json, os
out = "gpurun_out/r04b"
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, "attnio_" + c, "r04_counter_collection.csv")
    if not os.path.exists(p): continue
    per = {}
    for r in csv.DictReader(open(p)):
        if "attn_w4_kernel" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d = per.setdefault((k, int(r["Dispatch_Id"])), 0.0)
        per[(k, int(r["Dispatch_Id"]))] = d + float(r["Counter_Value"])
    agg = {}
    for (k, _), v in per.items(): agg.setdefault(k, []).append(v)
    for k, v in agg.items(): res.setdefault(k, {})[c + "_KB"] = sum(v) / len(v)
B, H, N = 8, 24, 4608
alg = 4 * B * N * H * 128 * 2
for k, e in res.items():
    e["algorithmic_bytes (q, k, v read once + o written)"] = alg
    if "FETCH_SIZE_KB" in e: e["fetch_bytes_corrected_x2"] = e["FETCH_SIZE_KB"] * 2048
    if "WRITE_SIZE_KB" in e: e["write_bytes"] = e["WRITE_SIZE_KB"] * 1024
json.dump({"note": "attn_w4_kernel at B=8 H=24 N=4608, mean of its launches in tools/pmc_attn.py; FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": res},
          open(os.path.join(out, "r04_attention_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $out/r04_gemm_pmc_traffic.json | head -60
