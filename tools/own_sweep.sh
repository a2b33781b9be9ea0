#!/bin/bash
# the algorithm-named kernels on the configurations that name them (smartgpu_tune(0,1)): bash tools/own_sweep.sh <tag>
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/sweep.py --corpus english --own --algos hor,bm,bndm --ms 2,4,8,16,32,64,128,256,512,1024,2048,4096 --reps 3 > $OUT/own_english.log 2>&1
python tools/sweep.py --sigma 4 --own --algos hor,bm,bndm,epsm --ms 2,4,8,16,32,64,256,4096 --reps 3 > $OUT/own_sigma4.log 2>&1
python tools/sweep.py --sigma 2 --own --algos hor,bm,bndm,epsm --ms 2,4,8,16,32,64,256,4096 --reps 3 > $OUT/own_sigma2.log 2>&1
python tools/sweep.py --sigma 128 --own --algos hor,bm,kmp,bndm --ms 4,8,16,32,64,256 --reps 3 > $OUT/own_sigma128.log 2>&1
python tools/sweep.py --corpus english --own --algos kmp --ms 8,16,32,128,1024 --reps 3 > $OUT/own_kmp_english.log 2>&1
python tools/sweep.py --sigma 2 --own --algos kmp --ms 8,16,32,128,1024 --reps 3 > $OUT/own_kmp_sigma2.log 2>&1
tail -n +2 $OUT/own_*.log | cut -c1-80
