#!/bin/bash
# The memory path of the four data paths (column / lane tiles: hor_scan; runs through LDS slabs: so_runs, kmp_runs; registers only:
# packed_scan), 1 GiB rand128: requests to memory by size, L2 hits, latency, stalls.   bash tools/pmc_memory_path.sh
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P2="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
P3="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
P4="TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum"
for spec in "hor 32" "so 32" "kmp 32" "epsm 4"; do
  set -- $spec
  EXTRA="--own" bash tools/pmc_probe.sh r04mem $1 $2 128 "$P1" "$P2" "$P3" "$P4" > gpurun_out/mem_$1.txt 2>&1 || exit 1
done
