#!/bin/bash
# One GPU-box session: parity tests, headline bench, rocprofv3 stats + PMC, sweep.
# Usage (from the repo root, on the GPU box): bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu" && timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "rc=$?"; tail -3 "$OUT/pytest_gpu.log"
echo "== sweep" && timeout -k 10 600 python tools/sweep.py --json "$OUT/sweep_rand128.json" > "$OUT/sweep_rand128.log" 2>&1; echo "rc=$?"; cat "$OUT/sweep_rand128.log"
echo "== bench" && timeout -k 10 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc=$?"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
cd /tmp
echo "== rocprofv3 stats" && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu > "$OUT/prof_stats.log" 2>&1; echo "rc=$?"
find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
echo "== rocprofv3 pmc FETCH_SIZE" && timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_pmc_fetch" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu > "$OUT/prof_pmc_fetch.log" 2>&1; echo "rc=$?"
echo "== rocprofv3 pmc WRITE_SIZE" && timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_pmc_write" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu > "$OUT/prof_pmc_write.log" 2>&1; echo "rc=$?"
cd "$ROOT"
find "$OUT" -name "*.csv" -size +2M -delete   # keep the merge under the 64 MiB cap
ls -la "$OUT" "$OUT"/prof_* 2>/dev/null | head -40
