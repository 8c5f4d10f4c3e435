#!/bin/bash
# scripts/variant.sh <name> "<reg ids>" <extra hipcc flags...> -- a timing-only variant of the library: the listed PLAN_REG translation
# units recompiled with the extra flags, everything else taken from the product build -> viterbidecodercpp_amd/libvit_hip_<name>.so
# (load it with VIT_HIP_LIB_PATH; scripts/gpu_ab.sh alternates two builds on one box)
set -e
NAME=$1; IDS=$2; shift 2
cd "$(dirname "$0")/../viterbidecodercpp_amd/csrc"
B=build_$NAME
mkdir -p $B
cp build/*.o $B/
for id in $IDS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function "$@" -DVIT_REG_ID=$id -c -o $B/reg_inst_$id.o reg_inst.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvit_hip_$NAME.so $B/*.o -ldl
echo built ../libvit_hip_$NAME.so
