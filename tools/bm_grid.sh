#!/bin/bash
# bm_scan: workgroup shape (smartgpu_tune(2,.)) x workgroups per CU (tune(4,.)), own kernel
run() { python tools/sweep.py $1 --own --algos bm --ms $2 --reps 3 --tune "$3" 2>&1 | grep "^bm" | awk -v t="$3" -v c="$1" '{printf "%-12s %-18s %-7s %s ms\n", t, c, $2, $4}'; }
for SH in 1 2; do for W in 3 4 5 6; do
  [ $SH = 2 ] && WW=$((W*2)) || WW=$W
  run "--sigma 128" 8,16,32,256 "2=$SH,4=$WW"
  run "--corpus english" 2,4,8,16,32,128,1024 "2=$SH,4=$WW"
  run "--sigma 4" 8,32 "2=$SH,4=$WW"
done; done
