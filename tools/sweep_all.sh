#!/bin/bash
# All the sweeps DESIGN.md §8 tabulates, on one GPU box: bash tools/sweep_all.sh <tag>
# (logs under gpurun_out/<tag>/; render with tools/tables.py).  *_own = smartgpu_tune(0,1): every algorithm on its
# own kernel, no rerouting by the plan.
set -o pipefail
TAG=${1:-sweeps}
OUT=$(pwd)/gpurun_out/$TAG
mkdir -p "$OUT"
MS_FULL=2,4,8,16,32,64,128,256,512,1024,2048,4096
run() { name=$1; shift; echo "== $name"; timeout -k 10 500 python tools/sweep.py "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -5 "$OUT/$name.log"; exit 1; }; tail -1 "$OUT/$name.log" | cut -c1-110; }
run sweep_rand128_full --ms $MS_FULL --reps 3 &&
run sweep_rand128_own --ms 4,8,16,32,64,256 --reps 3 --own &&
run sweep_rand4 --sigma 4 --ms 2,4,8,16,32,64 --reps 3 &&
run sweep_rand4_own --sigma 4 --ms 2,4,8,16,32,64 --reps 3 --own --algos hor,bm,bndm,epsm &&
run sweep_rand2 --sigma 2 --ms 2,4,8,16,32,64 --reps 3 &&
run sweep_rand2_own --sigma 2 --ms 2,4,8,16,32,64 --reps 3 --own --algos hor,bm,bndm,epsm &&
run sweep_english_4gib --corpus english --gib 4 --ms $MS_FULL --reps 3 &&
run sweep_english_4gib_own --corpus english --gib 4 --ms 2,4,8,16,32,64,256,1024,4096 --reps 3 --own --algos hor,bm,bndm &&
run sweep_cfg5_rand2_4gib --sigma 2 --gib 4 --ms $MS_FULL --algos hor,bm,kmp,so,epsm --reps 3 &&
run sweep_cfg5_rand32_4gib --sigma 32 --gib 4 --ms $MS_FULL --algos hor,bm,kmp,so,epsm --reps 3 &&
run sweep_cfg5_rand256_4gib --sigma 256 --gib 4 --ms $MS_FULL --algos hor,bm,kmp,so,epsm --reps 3 &&
F3=sa,qs,tunedbm,raita,hash3,hash5,hash8,sbndm,kr,bndml &&
run sweep_f3_rand128 --algos $F3 --ms 8,16,32,64,128,256,512,1024,2048,4096 --reps 3 &&
run sweep_f3_rand4 --algos $F3 --sigma 4 --ms 8,16,32,64 --reps 3 &&
run sweep_f3_english_4gib --algos $F3 --corpus english --gib 4 --ms 8,16,32,64,256,1024,4096 --reps 3
