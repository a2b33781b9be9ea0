#!/bin/bash
# A/B of the runs kernels on one GPU box: new so_runs / kmp_runs against the previous ones.
# Usage: bash tools/ab_runs.sh <tag>
set -o pipefail
TAG=${1:-ab}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 python tools/sweep.py "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -5 "$OUT/$name.log"; return 1; }; grep -v "^streaming" "$OUT/$name.log" | awk '{print $1,$2,$3,$4,$5,$10,$11}' ; }
MS=2,4,8,16,29,30,32,33,64,256,1024,4096
run new_rand128 --algos kmp,so,sa,epsm --ms $MS --reps 3 &&
run old_rand128 --algos kmp,so,sa --ms $MS --reps 3 --tune 3=3,6=4 &&
run new_rand4 --algos kmp,so,epsm --sigma 4 --ms 2,4,8,16,32,64 --reps 3 &&
run old_rand4 --algos kmp,so --sigma 4 --ms 2,4,8,16,32,64 --reps 3 --tune 3=3,6=4 &&
run new_rand2 --algos kmp,so,hor,epsm --sigma 2 --ms 2,4,8,16,32,64 --reps 3 &&
run old_rand2 --algos kmp,so,hor --sigma 2 --ms 2,4,8,16,32,64 --reps 3 --tune 3=3,6=4
