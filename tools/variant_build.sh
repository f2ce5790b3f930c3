#!/bin/bash
# build a variant of nm_features.hip into build_abl/lib_<name>.so (the other objects are the in-tree ones):
#   bash tools/variant_build.sh <name> [hipcc flags / -D defines for nm_features.hip]
#   SRC=nm_index bash tools/variant_build.sh <name> [flags]        (another source file instead)
#   NIMRUD_HIP_LIBRARY=$PWD/build_abl/lib_<name>.so python bench.py ...
set -e
NAME=$1; shift
SRC=${SRC:-nm_features}
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/nimrud_amd/csrc
mkdir -p "$R/build_abl"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$C/$SRC.hip" -o "$R/build_abl/${SRC}_$NAME.o" 2> "$R/build_abl/$NAME.remarks" || { tail -20 "$R/build_abl/$NAME.remarks"; exit 1; }
grep -A12 "k_scale_featuresILi7ELb1ELb0ELb1" "$R/build_abl/$NAME.remarks" | grep -E "VGPRs:|ScratchSize" | head -2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/build_abl/lib_$NAME.so" \
    $(for f in nm_api nm_index nm_halo nm_field nm_features; do [ $f = $SRC ] || echo "$C/$f.o"; done) "$R/build_abl/${SRC}_$NAME.o" \
    -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -f "$R/build_abl/${SRC}_$NAME.o"
