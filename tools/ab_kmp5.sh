#!/bin/bash
# kmp_runs, COMPACT (five four-wave workgroups per CU, round 4) against round 3's one-workgroup form (tune(3,6), A/B build),
# alternating in one call: bash tools/ab_kmp5.sh "<ms>" [more sweep args, e.g. --sigma 32 | --corpus english]
MS=$1; shift
for round in 1 2; do
  for t in "" "3=6"; do
    echo "== tune [$t] round $round"
    SMARTGPU_LIB=smart_amd/csrc/libsmartgpu_ab.so python tools/sweep.py --algos kmp,so --ms $MS --reps 5 --own --tune "$t" "$@" 2>&1 | grep -v "^streaming" | awk '{printf "%-6s %-7s %-10s %s ms  %s%%  %s\n", $1,$2,$3,$4,$11,$NF}'
  done
done
