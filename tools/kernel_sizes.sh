#!/bin/bash
# Code size (bytes) of every kernel of smart_amd/csrc/k_*.hip: bash tools/kernel_sizes.sh [extra hipcc flags]
# (the instruction cache is 64 KB per two CUs; sixteen waves spread over a loop larger than that miss in it)
set -e
T=$(mktemp -d)
for f in "$(dirname "$0")"/../smart_amd/csrc/k_*.hip; do
  u=$(basename $f .hip); [ $u = k_ab ] && continue
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only "$@" -c -o $T/$u.o $f
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/$u.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$u.co
  /opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/$u.co | awk '$4=="FUNC"{print $3, $8}' | sort -n | c++filt | cut -c1-120
done
rm -rf $T
