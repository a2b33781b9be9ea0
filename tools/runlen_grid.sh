#!/bin/bash
# Run length of the runs kernels (smartgpu_tune(5, bytes); default 2048: runs of 2-4 KiB), one binary, alternating
for round in 1 2; do for L in 2048 1024 4096 8192; do echo "== tune 5=$L round $round"
  python tools/sweep.py --sigma 4 --algos so,kmp,epsm --ms 32,1024 --reps 5 --tune 5=$L | grep "^so\|^kmp" | cut -c1-92
  python tools/sweep.py --sigma 128 --algos so,kmp,epsm --ms 32,1024 --reps 5 --tune 5=$L | grep "^so\|^kmp" | cut -c1-92
  python tools/sweep.py --corpus english --algos so,kmp,epsm --ms 4,32,1024 --reps 5 --tune 5=$L | grep "^so\|^kmp" | cut -c1-92
done; done
