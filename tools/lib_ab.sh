#!/bin/bash
# In-box A/B of variant libraries (tools/build_variant.sh) on the 57-block DiT forward (B = 8, 1024 x 1024; tools/dit_ab.py), each library in
# its own process, the whole list repeated ROUNDS times (alternating: box drift shows up as a spread inside a name, not as a difference).
#   bash tools/lib_ab.sh <out.log> <rounds> [--fp8] name [name ...]      ("default" = textflux_amd/libtextflux_hip.so)
out=$1; rounds=$2; shift 2
fp8=""; if [ "$1" = "--fp8" ]; then fp8="--fp8"; shift; fi
mkdir -p $(dirname $out)
for r in $(seq 1 $rounds); do
  for n in "$@"; do
    if [ $n = default ]; then unset TFX_LIB; else export TFX_LIB=$PWD/textflux_amd/libtextflux_hip_exp_$n.so; [ -f $TFX_LIB ] || { echo "missing $n" >> $out; continue; }; fi
    echo "== $n $fp8 (round $r)" >> $out
    timeout 400 python tools/dit_ab.py $fp8 2>&1 | grep "ms/forward" >> $out
  done
done
unset TFX_LIB
cat $out
