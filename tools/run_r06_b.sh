#!/bin/bash
# round 6, second GPU call: the suite on the round-6 epilogue (packed q/k-norm pre-pass, v_dot2 weighting, table by buffer loads; fp8 q/k-norm
# kernel without the spilled head), then library A/Bs on the DiT forward (bf16 and fp8), the shape table and the q-side A/B per library.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06d; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 ) > $out/gputests.log 2>&1
tail -25 $out/gputests.log
bash tools/lib_ab.sh $out/ab_bf16.log 3 r5gemm default > /dev/null 2>&1; cat $out/ab_bf16.log
bash tools/lib_ab.sh $out/ab_fp8.log 2 --fp8 r5gemm default > /dev/null 2>&1; cat $out/ab_fp8.log
timeout 600 python tools/gemm_shapes_power.py --tag r06 --hipblaslt --out $out/r06_gemm_shapes.jsonl > $out/shapes.log 2>&1; tail -2 $out/shapes.log
timeout 300 python tools/qkn_ab6.py $out/r06_qkn_ab.json > $out/qkn_ab.log 2>&1; grep -c tflops $out/qkn_ab.log
TFX_LIB=$PWD/textflux_amd/libtextflux_hip_exp_r5gemm.so timeout 300 python tools/qkn_ab6.py $out/r06_qkn_ab_r5gemm.json > $out/qkn_ab_r5.log 2>&1
python - <<'PY'
import json
for f in ("r06_qkn_ab.json", "r06_qkn_ab_r5gemm.json"):
    try:
        rows = json.load(open("gpurun_out/r06b/" + f))
        print(f)
        for r in rows: print("  ", r["round"], r["name"][:50].ljust(50), r["ms_per_launch"], r["tflops"], r["sclk_mhz"], r["board_w"])
    except Exception as e: print(f, e)
PY
ls $out
