#!/bin/bash
# In-call A/B of library variants (tools/build_variant.sh): bash tools/ab_libs.sh "<sweep args>" <suffix> <suffix> ...
# ("-" = the product library); every variant is run twice, in alternating order (boxes differ by several percent).
ARGS=$1; shift
for round in 1 2; do
  for v in "$@"; do
    lib=smart_amd/csrc/libsmartgpu_$v.so; [ "$v" = "-" ] && lib=smart_amd/csrc/libsmartgpu.so
    echo "== [$v] round $round"
    SMARTGPU_LIB=$lib python tools/sweep.py $ARGS 2>&1 | grep -v "^streaming" | awk '{printf "%-6s %-7s %-10s %s ms  %s%%\n", $1,$2,$3,$4,$11}'
  done
done
