#!/bin/bash
# bndm_scan: workgroup shape x workgroups per CU x q on small alphabets (own kernel)
run() { python tools/sweep.py $1 --own --algos bndm --ms $2 --reps 3 --tune "$3" 2>&1 | grep "^bndm" | awk -v t="$3" '{printf "%-22s %s %-7s %s ms %s%%\n", t, $3, $2, $4, $11}'; }
for SH in 1 2; do for W in 4 5 6 8; do
  [ $SH = 2 ] && WW=$((W*2)) || WW=$W
  run "--sigma 4" 8,16,32 "2=$SH,4=$WW"
  run "--sigma 2" 16,32 "2=$SH,4=$WW"
done; done
for Q in 2 4 8; do run "--sigma 4" 8,16,32 "2=1,4=5,1=$Q"; run "--sigma 2" 16,32 "2=1,4=5,1=$Q"; done
