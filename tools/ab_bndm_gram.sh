#!/bin/bash
# bndm_scan with gram tables (texts of at most four byte values, round 4) against its mask loop (tune(1,9)), own kernel,
# alternating in one call: bash tools/ab_bndm_gram.sh <sigma> "<ms>"
SIGMA=$1; MS=$2
for round in 1 2; do
  for t in "" "1=9"; do
    echo "== sigma $SIGMA tune [$t] round $round"
    python tools/sweep.py --algos bndm,so --sigma $SIGMA --ms $MS --reps 3 --own --tune "$t" 2>&1 | grep -v "^streaming" | awk '{printf "%-6s %-7s %-10s %s ms  %s%%  %s\n", $1,$2,$3,$4,$11,$NF}'
  done
done
