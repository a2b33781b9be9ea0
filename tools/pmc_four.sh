#!/bin/bash
# The runs kernels on a four-symbol text, with and without their four-bytes-per-step tables: bash tools/pmc_four.sh <tag>
TAG=$1
CTR=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE")
for T in "3=5" "3=0"; do EXTRA="--tune $T" bash tools/pmc_probe.sh ${TAG}_kmp_$(echo $T | tr = _) kmp 32 4 "${CTR[@]}" > gpurun_out/${TAG}_kmp_$(echo $T | tr = _).txt 2>&1; done
for T in "6=5" "6=0"; do EXTRA="--tune $T" bash tools/pmc_probe.sh ${TAG}_so_$(echo $T | tr = _) so 32 4 "${CTR[@]}" > gpurun_out/${TAG}_so_$(echo $T | tr = _).txt 2>&1; done
tail -n 20 gpurun_out/${TAG}_*.txt
