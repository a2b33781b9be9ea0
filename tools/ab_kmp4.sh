#!/bin/bash
# kmp_runs on small alphabets with and without the four-bytes-per-step table (tune(3,5)), alternating
for round in 1 2; do for T in "3=5" "3=0"; do echo "== tune $T round $round";
  python tools/sweep.py --sigma 2 --algos kmp,epsm --ms 9,16,32,64,128,256,4096 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
  python tools/sweep.py --sigma 4 --algos kmp,epsm --ms 9,16,32,64,1024 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
  python tools/sweep.py --sigma 2 --own --algos kmp,epsm --ms 2,4,8 --reps 3 --tune $T | grep "^kmp\|MISMATCH" | cut -c1-92
done; done
