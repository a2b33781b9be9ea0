#!/bin/bash
# bndm_scan on the configurations that name it, product library against a baseline build: bash tools/ab_bndm.sh <suffix of the baseline lib>
for v in "$@" -; do
  lib=smart_amd/csrc/libsmartgpu_$v.so; [ "$v" = "-" ] && lib=smart_amd/csrc/libsmartgpu.so
  echo "== [$v]"
  for S in 4 2 128; do SMARTGPU_LIB=$lib python tools/sweep.py --sigma $S --own --algos bndm --ms 2,4,8,16,32,64 --reps 3 2>&1 | grep "^bndm" | cut -c1-86; done
  SMARTGPU_LIB=$lib python tools/sweep.py --corpus english --own --algos bndm --ms 4,8,16,32,64,256,1024 --reps 3 2>&1 | grep "^bndm" | cut -c1-86
done
