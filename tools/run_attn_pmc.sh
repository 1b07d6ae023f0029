#!/bin/bash
# PMC passes over the two one-wave-per-SIMD attention kernels (options 30 and 40).  usage: bash tools/run_attn_pmc.sh ; gpurun_out/attn_pmc/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/attn_pmc; mkdir -p $out
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $out/p$i -o t --output-format csv -- python tools/pmc_attn.py > $out/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, json
out = "gpurun_out/attn_pmc"
res = {}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "attn_w" in n:
            a = acc.setdefault((n, r["Counter_Name"]), [0.0, set()])
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for (n, c), (v, ds) in acc.items():
        res.setdefault(n, {})[c] = v / max(len(ds), 1)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for n, e in res.items():
    print(n); print("   ", {k: round(v) for k, v in sorted(e.items())})
    if "SQ_WAVE_CYCLES" in e:
        w = e["SQ_WAVE_CYCLES"]
        print("    shares of wave cycles:", {k: round(e[k] / w, 3) for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if k in e})
    if "GRBM_GUI_ACTIVE" in e and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        print("    mfma util", round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024), 3))
PY
