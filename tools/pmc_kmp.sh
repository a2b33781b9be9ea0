#!/bin/bash
# kmp_runs PMC passes with and without the speculative halves: bash tools/pmc_kmp.sh <tag>
TAG=$1
for T in "3=4" "3=0"; do
  EXTRA="--tune $T" bash tools/pmc_probe.sh ${TAG}_$(echo $T | tr = _) kmp 32 128 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" > /dev/null 2>&1
done
