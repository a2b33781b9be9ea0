#!/bin/bash
# kmp_runs with and without its speculation (tune(3,4)), alternating: bash tools/ab_kmp.sh
for round in 1 2; do for T in "3=4" "3=0"; do echo "== tune $T round $round";
  python tools/sweep.py --corpus english --algos kmp,epsm --ms 9,16,32,64,256,1024 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
  python tools/sweep.py --sigma 32 --algos kmp,epsm --ms 16,64,1024 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
  python tools/sweep.py --sigma 16 --algos kmp,epsm --ms 16,64,1024 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
  python tools/sweep.py --sigma 8 --algos kmp,epsm --ms 16,64 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
  python tools/sweep.py --algos kmp,epsm --ms 16,256 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
done; done
