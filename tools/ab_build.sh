#!/bin/bash
# A/B library: one csrc file rebuilt with extra flags, linked with the other objects of the product build -> exp/lib_<tag>.so
# usage: tools/ab_build.sh <tag> <file.hip> [extra hipcc flags...]   then APAD_LIB_PATH=exp/lib_<tag>.so python ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/ap-adapter_amd/csrc; tag=$1; f=$2; shift 2
mkdir -p $R/exp
extra=""
vg="-mllvm -amdgpu-mfma-vgpr-form"; [ "$f" = mlp.hip ] && vg="-mllvm -amdgpu-use-amdgpu-trackers=1"; { [ "$f" = mlp3.hip ] || [ "$f" = geglu3.hip ]; } && vg=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $vg $extra "$@" -c $C/$f -o $R/exp/ab_$tag.o
objs=$(ls $C/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/exp/lib_$tag.so $objs $R/exp/ab_$tag.o
echo built exp/lib_$tag.so
