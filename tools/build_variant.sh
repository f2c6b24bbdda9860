#!/bin/bash
# Development aid: an experimental variant of the library (extra -D flags on ONE source; the other objects come from the regular build).
# usage (build container): bash tools/build_variant.sh <name> <source.hip> [-DFLAG ...]   ->  chinesecheckersagent_amd/libccsp_exp_<name>.so
# (the .so travels with the gpurun snapshot; use it through CCSP_LIB=$PWD/chinesecheckersagent_amd/libccsp_exp_<name>.so)
set -e
name=$1; src=$2; shift 2
P=chinesecheckersagent_amd
python -m chinesecheckersagent_amd.build >/dev/null
extra=""
[ "$src" = ccsp_net.hip ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 $extra "$@" -c $P/csrc/$src -o /tmp/variant_$name.o
objs=""
for f in ccsp_rules_kernels ccsp_engine ccsp_net ccsp_host; do
  if [ "$f.hip" = "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $P/build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -fPIC -shared -o $P/libccsp_exp_$name.so $objs
echo $P/libccsp_exp_$name.so
