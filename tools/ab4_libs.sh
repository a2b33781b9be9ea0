#!/bin/bash
# four alternating rounds of two libs
ARGS=$1; A=$2; B=$3
for round in 1 2 3 4; do for v in $A $B; do
  lib=smart_amd/csrc/libsmartgpu_$v.so; [ "$v" = "-" ] && lib=smart_amd/csrc/libsmartgpu.so
  echo "== [$v] round $round"
  SMARTGPU_LIB=$lib python tools/sweep.py $ARGS 2>&1 | grep -v "^streaming" | awk '{printf "%-6s %-7s %-10s %s ms  %s%%\n", $1,$2,$3,$4,$11}'
done; done
