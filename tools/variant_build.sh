#!/bin/bash
# build the WORKING TREE's library with extra flags into build_abl/lib_<name>.so for same-box A/B runs:
#   bash tools/variant_build.sh <name> [-DNM_... flags]
#   NIMRUD_HIP_LIBRARY=$PWD/build_abl/lib_<name>.so python bench.py ...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build_abl/obj_$NAME
rm -rf "$D"; mkdir -p "$D"
cd "$R/nimrud_amd/csrc"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math"
for f in *.hip; do
  EXTRA=""
  [ "$f" = nm_features.hip ] && EXTRA="-mllvm -disable-machine-licm -fconstexpr-steps=20000000 -mllvm -amdgpu-sched-strategy=max-memory-clause"
  /opt/rocm/bin/hipcc $BASE $EXTRA "$@" -c $f -o "$D/${f%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build_abl/lib_$NAME.so" "$D"/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$D"
ls -la "$R/build_abl/lib_$NAME.so"
