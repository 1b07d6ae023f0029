#!/bin/bash
# Where does hipBLASLt's kernel differ from ours on the dominant GEMM shape?  Kernel trace (name = Tensile config, registers, LDS,
# workgroup size) + separate --pmc passes.  usage: bash tools/run_gemm_vs_blaslt.sh ; results gpurun_out/blaslt/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/blaslt; mkdir -p $out
rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python tools/pmc_gemm_vs_blaslt.py > $out/trace.log 2>&1
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o t --output-format csv -- python tools/pmc_gemm_vs_blaslt.py > $out/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, json, os
out = "gpurun_out/blaslt"
res = {}
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm8pp" in n or "Cijk" in n or "MT" in n:
            e = res.setdefault(n[:400], {"n": 0, "ns": 0})
            e["n"] += 1; e["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for k in ("Workgroup_Size", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "Grid_Size", "Grid_Size_X"):
                if k in r: e[k] = r[k]
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm8pp" in n or "Cijk" in n or "MT" in n:
            a = acc.setdefault((n[:400], r["Counter_Name"]), [0.0, set()])
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for (n, c), (v, ds) in acc.items():
        res.setdefault(n, {})[c] = v / max(len(ds), 1)
for n, e in res.items():
    if e.get("n"): e["avg_us"] = e["ns"] / e["n"] / 1e3
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for n, e in res.items():
    print(n[:300]); print("   ", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()})
PY
