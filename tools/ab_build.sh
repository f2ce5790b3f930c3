#!/bin/bash
# build the library of a git revision into build_abl/lib_<name>.so for same-box A/B runs:
#   bash tools/ab_build.sh <rev> <name> [extra hipcc flags]
#   NIMRUD_HIP_LIBRARY=$PWD/build_abl/lib_<name>.so python bench.py ...
set -e
REV=$1; NAME=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build_abl/$NAME
rm -rf "$D"; mkdir -p "$D/nimrud_amd/csrc" "$D/include"
for f in $(git -C "$R" ls-tree --name-only "$REV" nimrud_amd/csrc/ | grep -E '\.(hip|h)$'); do
  git -C "$R" show "$REV:$f" > "$D/$f"
done
git -C "$R" show "$REV:include/nimrud_hip.h" > "$D/include/nimrud_hip.h"
cd "$D/nimrud_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build_abl/lib_$NAME.so" *.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$D"
ls -la "$R/build_abl/lib_$NAME.so"
