"""Aggregate a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass over bench.py into per-kernel and
whole-DiT matrix-pipe utilisation.

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the busy counter is summed over the 1024 SIMDs
(= 32 cycles per v_mfma_f32_32x32x16_bf16, checked against 2*M*N*K / 32768 of a known GEMM), GUI_ACTIVE over the 8 XCDs.
usage: python tools/pmc_bench_util.py <counter_collection.csv> <out.json>"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
disp = {}
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = {}
for d in disp.values():
    n = d["name"]
    key = n.split("(")[0].replace("void ", "")
    a = agg.setdefault(key, {"launches": 0, "mfma_busy": 0.0, "gui_active": 0.0})
    a["launches"] += 1
    a["mfma_busy"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a["gui_active"] += d.get("GRBM_GUI_ACTIVE", 0.0)
def util(a):
    return a["mfma_busy"] / (a["gui_active"] / 8 * 1024) if a["gui_active"] else 0.0
out = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", "kernels": {}}
dit = {"launches": 0, "mfma_busy": 0.0, "gui_active": 0.0}
tot = {"launches": 0, "mfma_busy": 0.0, "gui_active": 0.0}
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["gui_active"]):
    share = a["gui_active"]
    is_dit = k.startswith("tfx::") and any(n in k for n in ("gemm8pp_kernel", "attn_w4_kernel", "attn_w16_kernel", "attn_mx_kernel", "attn_kernel", "ln_modulate_kernel",
                                                             "rmsnorm_rope_kernel", "sched_step_kernel", "splitk_reduce_kernel", "tail_reduce_kernel", "quant_rows_fp8"))
    for t in (tot,) + ((dit,) if is_dit else ()):
        for f in ("launches", "mfma_busy", "gui_active"):
            t[f] += a[f]
    out["kernels"][k] = {"launches": a["launches"], "gpu_cycles_per_xcd": a["gui_active"] / 8, "mfma_util": round(util(a), 4)}
out["dit_kernels_total"] = {"launches": dit["launches"], "mfma_util": round(util(dit), 4),
                            "share_of_all_gpu_cycles": round(dit["gui_active"] / tot["gui_active"], 4)}
out["all_kernels_total"] = {"launches": tot["launches"], "mfma_util": round(util(tot), 4)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for k, v in list(out["kernels"].items())[:10]:
    print(k[:70], v)
