#!/bin/bash
# PMC probe of one (algo, m, sigma): [EXTRA="--corpus english --own"] bash tools/pmc_probe.sh <tag> <algo> <m> <sigma> "<counters pass 1>" "<counters pass 2>" ...
set -o pipefail
TAG=$1; ALGO=$2; M=$3; SIGMA=$4; shift 4
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc_${ALGO}_m${M}_s${SIGMA}_p$i" -- python3 "$ROOT/tools/sweep.py" --algos $ALGO --ms $M --sigma $SIGMA --reps 3 $EXTRA > "$OUT/pmc_${ALGO}_m${M}_p$i.log" 2>&1 || echo "pass $i failed"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for f in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "_scan" in r["Kernel_Name"] or "_runs" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-20:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()): print("%-22s %-28s n=%d mean=%.1f"%(k[0],k[1],len(v),sum(v)/len(v)))
PY
find "$OUT" -name "*.csv" -size +2M -delete
