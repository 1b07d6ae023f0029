#!/bin/bash
# A/B of cache-policy hints (nt = non-temporal) on the DiT's write-once / read-once streams: variant libraries built by hand
# (textflux_amd/libtextflux_hip_exp_<name>.so: -DTFX_GSTORE_AUX=2 GEMM output stores, -DTFX_GRES_AUX=2 residual reads, -DTFX_GLDS_AUX=2 operand requests,
# -DTFX_ATT_O_NT attention output stores, -DTFX_LN_ST_NT / -DTFX_LN_LD_NT LayerNorm+modulation stores / loads), each timed on the 57-block DiT forward
# (B = 8, 1024 x 1024) in its own process on ONE box; usage: bash tools/run_r04_cachepolicy.sh name [name ...]
mkdir -p gpurun_out
out=gpurun_out/r4_cachepolicy.log
for n in "$@"; do
  if [ $n = default ]; then unset TFX_LIB; else export TFX_LIB=$PWD/textflux_amd/libtextflux_hip_exp_$n.so; [ -f $TFX_LIB ] || continue; fi
  echo "== $n" >> $out
  timeout 300 python tools/dit_ab.py 2>&1 | grep "ms/forward" >> $out
done
cat $out
