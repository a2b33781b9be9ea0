#!/bin/bash
# One GPU-box session: parity tests, headline bench, rocprofv3 stats + PMC passes.
# Usage (from the repo root, on the GPU box): bash tools/gpu_round.sh <tag>
# The session stops at the first failing step: profiles of a build that fails parity are not collected.
set -o pipefail
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { echo "== $1"; shift; "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "FAILED rc=$rc: session $TAG stops here"; echo failed > "$OUT/SESSION_FAILED"; exit $rc; fi; }
rm -f "$OUT/SESSION_FAILED"
python3 -c "import json; from smart_amd import sources; print(json.dumps(sources.all_unit_shas()))" > "$OUT/kernel_sources.sha256.json"   # what roofline.traffic gets bound to, per kernel family (collect_profiles.py)
step "pytest -m gpu" bash -c "timeout -k 10 900 python -m pytest tests -m gpu -x -q > '$OUT/pytest_gpu.log' 2>&1; rc=\$?; tail -3 '$OUT/pytest_gpu.log'; exit \$rc"
step "bench" bash -c "timeout -k 10 900 python bench.py --sweep-out '$OUT/bench_sweep.json' > '$OUT/bench.json' 2> '$OUT/bench.err'; rc=\$?; cut -c1-600 '$OUT/bench.json'; tail -3 '$OUT/bench.err'; exit \$rc"
cd /tmp
step "rocprofv3 stats" bash -c "timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d '$OUT/prof_stats' -- python3 '$ROOT/bench.py' --steps 20 --warmup 5 --no-cpu --no-sweep > '$OUT/prof_stats.log' 2>&1"
find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1 | xargs -r head -8
pmc() {  # pmc <name> <bench args...>: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
  NAME=$1; shift
  step "rocprofv3 pmc FETCH_SIZE $NAME" bash -c "timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d '$OUT/prof_pmc_fetch_$NAME' -- python3 '$ROOT/bench.py' $* --steps 5 --warmup 2 --no-cpu --no-sweep > '$OUT/prof_pmc_fetch_$NAME.log' 2>&1"
  step "rocprofv3 pmc WRITE_SIZE $NAME" bash -c "timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d '$OUT/prof_pmc_write_$NAME' -- python3 '$ROOT/bench.py' $* --steps 5 --warmup 2 --no-cpu --no-sweep > '$OUT/prof_pmc_write_$NAME.log' 2>&1"
}
for ALGO in hor kmp so epsm; do pmc $ALGO --algo $ALGO; done
# the serial and packed kernels on the other kinds of text: a binary alphabet, the English corpus
for ALGO in kmp so epsm; do pmc $ALGO-sigma2 --algo $ALGO --sigma 2; pmc $ALGO-english --algo $ALGO --corpus english; done
cd "$ROOT"
find "$OUT" -name "*.csv" -size +2M -delete   # keep the merge under the 64 MiB cap
ls "$OUT"
