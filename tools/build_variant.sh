#!/bin/bash
# Builds a variant product library textflux_amd/libtextflux_hip_exp_<name>.so for in-box A/Bs (git-ignored, shipped by gpurun):
#   tools/build_variant.sh <name> "<extra hipcc flags>" [file=override.hip ...]
# every csrc object is compiled into /tmp/tfx_var_<name>/ with the product flags plus the extra ones; `file=path` takes that source instead of
# csrc/<file> (e.g. gemm.hip=/tmp/old/gemm.hip: a previous commit's kernel against today's other files).
set -e
name=$1; extra=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); src=$root/textflux_amd/csrc; bd=/tmp/tfx_var_$name; mkdir -p $bd
declare -A ov; for kv in "$@"; do ov[${kv%%=*}]=${kv#*=}; done
objs=""
for f in elementwise.hip imageops.hip gemm.hip attention.hip attention_w4.hip textenc.hip capi.cpp launch.cpp; do
  s=${ov[$f]:-$src/$f}; o=$bd/${f%.*}.o; fl=""
  [ $f = attention_w4.hip ] && fl="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I$src -I$root/include $fl $extra -c $s -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $root/textflux_amd/libtextflux_hip_exp_$name.so
ls -la $root/textflux_amd/libtextflux_hip_exp_$name.so
