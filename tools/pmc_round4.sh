#!/bin/bash
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
EXTRA="--own" bash tools/pmc_probe.sh r04pmc epsm 4 128 "$P1" "$P2" "$P3" > gpurun_out/pmc_epsm_m4_rand128.txt 2>&1 &&
EXTRA="--own" bash tools/pmc_probe.sh r04pmc hor 32 2 "$P1" "$P2" "$P3" > gpurun_out/pmc_hor_gram_m32_rand2.txt 2>&1 &&
EXTRA="--own" bash tools/pmc_probe.sh r04pmc bndm 8 4 "$P1" "$P2" "$P3" > gpurun_out/pmc_bndm_gram_m8_rand4.txt 2>&1 &&
EXTRA="--own" bash tools/pmc_probe.sh r04pmc kmp 32 128 "$P1" "$P2" "$P3" > gpurun_out/pmc_kmp_compact_m32_rand128.txt 2>&1 &&
EXTRA="--own" bash tools/pmc_probe.sh r04pmc hor 32 128 "$P1" "$P2" "$P3" > gpurun_out/pmc_hor_m32_rand128.txt 2>&1
tail -n 14 gpurun_out/pmc_*.txt
