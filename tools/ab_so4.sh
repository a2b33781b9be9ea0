#!/bin/bash
# so_runs on texts of at most four symbols with and without the four-symbol table (tune(6,5)), alternating;
# the last block: the same against another build (tools/ab_libs.sh), which also sees what the new code costs elsewhere
for round in 1 2; do for T in "6=5" "6=0"; do echo "== tune $T round $round";
  python tools/sweep.py --sigma 2 --algos so,epsm --ms 2,4,8,16,32,64,1024 --reps 3 --tune $T | grep "^so\|MISMATCH" | cut -c1-92
  python tools/sweep.py --sigma 4 --algos so,epsm --ms 2,4,8,16,32,64 --reps 3 --tune $T | grep "^so\|MISMATCH" | cut -c1-92
done; done
