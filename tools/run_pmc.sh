#!/bin/bash
# usage: tools/run_pmc.sh <outname> <script.py> COUNTER [COUNTER...]   (one PMC pass, counters only + kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; script=$2; shift 2
mkdir -p gpurun_out/prof
rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/prof/$name -o r01 --output-format csv -- python $script > gpurun_out/prof/$name.log 2>&1
tail -2 gpurun_out/prof/$name.log
python - "$name" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/prof/{sys.argv[1]}/r01_counter_collection.csv")))
agg = {}
for r in rows:
    n = r["Kernel_Name"][:44]
    if "tfx" not in n:
        continue
    agg.setdefault((n, int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: kv[0][1]):
    print(k, {a: int(b) for a, b in v.items()})
PY
