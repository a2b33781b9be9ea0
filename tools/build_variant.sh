#!/bin/bash
# Build an experimental variant of the library next to the product one:
#   bash tools/build_variant.sh <name> "<extra hipcc flags>"   ->  smart_amd/csrc/libsmartgpu_<name>.so
# Run with SMARTGPU_LIB=smart_amd/csrc/libsmartgpu_<name>.so python tools/sweep.py ...
set -e
NAME=$1; shift
D=$(cd "$(dirname "$0")/../smart_amd/csrc" && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value "$@" --offload-arch=gfx950 -c -o $T/kernels.o $D/kernels.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value "$@" --offload-arch=gfx950 -c -o $T/api.o $D/api.cpp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -c -o $T/tables.o $D/tables.cpp
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $D/libsmartgpu_$NAME.so $T/kernels.o $T/api.o $T/tables.o -ldl
rm -rf $T
echo built $D/libsmartgpu_$NAME.so
