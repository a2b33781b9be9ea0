#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   bash tools/build_variant.sh <name> "<extra hipcc flags>"   ->  smart_amd/csrc/libsmartgpu_<name>.so
# Run with SMARTGPU_LIB=smart_amd/csrc/libsmartgpu_<name>.so python tools/sweep.py ...
# (every unit of the product build — one per kernel family — compiled with the extra flags into a scratch directory)
set -e
NAME=$1; shift
D=$(cd "$(dirname "$0")/../smart_amd/csrc" && pwd)
T=$(mktemp -d)
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value"
for u in k_hor k_horg k_bm k_bmg k_bndm k_bndmx k_so k_kmp k_packed k_util launch; do
  /opt/rocm/bin/hipcc $FLAGS "$@" --offload-arch=gfx950 -c -o $T/$u.o $D/$u.hip &
done
/opt/rocm/bin/hipcc $FLAGS "$@" --offload-arch=gfx950 -c -o $T/api.o $D/api.cpp &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC "$@" -c -o $T/tables.o $D/tables.cpp &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libsmartgpu_$NAME.so $T/*.o -ldl
rm -rf $T
echo built $D/libsmartgpu_$NAME.so
