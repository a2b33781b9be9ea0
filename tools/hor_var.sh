#!/bin/bash
# hor_scan on patterns whose symbols repeat: the flat form (default) against the nested loop (smartgpu_tune(2,3)), alternating
for round in 1 2; do for T in "2=0" "2=3"; do echo "== tune $T round $round"
python tools/sweep.py --sigma 32 --own --algos hor,epsm --ms 8,16,32 --reps 5 --tune $T | grep "^hor" | cut -c1-90
python tools/sweep.py --corpus english --own --algos hor,epsm --ms 8,16,32 --reps 5 --tune $T | grep "^hor" | cut -c1-90
python tools/sweep.py --sigma 256 --own --algos hor,epsm --ms 8,16 --reps 5 --tune $T | grep "^hor" | cut -c1-90
done; done
