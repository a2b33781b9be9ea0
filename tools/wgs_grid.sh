#!/bin/bash
# Workgroups per CU of the flat skip loops where windows survive (hor_scan VAR 9, bm_scan; 128 threads each): smartgpu_tune(4, W)
run() { python tools/sweep.py $1 --own --algos hor,bm --ms $2 --reps 5 --tune "4=$3" 2>&1 | grep "^bm\|^hor" | awk -v t="$3" -v c="$1" '{printf "wgs=%-3s %-18s %-5s %-7s %s ms\n", t, c, $1, $2, $4}'; }
for round in 1 2; do for W in 12 10 13 14; do
  run "--corpus english" 8,32,128,1024 $W
  run "--sigma 32" 8,32 $W
done; done
