#!/usr/bin/env python3
"""Copy the judged summaries of one tools/gpu_round.sh session from gpurun_out/<tag>/ (scratch)
into profiles/<round>/<prefix>_* (tracked) and refresh profiles/pmc_traffic.json.

    python tools/collect_profiles.py r01n r01 n

Writes: <prefix>_bench_default.json, <prefix>_bench_hor_m32_kernel_stats.csv (rocprofv3 --stats),
<prefix>_bench_hor_m32_pmc_summary.csv (FETCH_SIZE / WRITE_SIZE per kernel, averaged over
dispatches, KiB as reported), <prefix>_sweep_rand128_1gib.log, <prefix>_pytest_gpu.log.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def pmc_rows(root, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, rnd, prefix = sys.argv[1:4]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(here, "gpurun_out", tag)
    dst = os.path.join(here, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out = lambda name: os.path.join(dst, f"{prefix}_{name}")

    shutil.copy(os.path.join(src, "bench.json"), out("bench_default.json"))
    shutil.copy(os.path.join(src, "sweep_rand128.log"), out("sweep_rand128_1gib.log"))
    shutil.copy(os.path.join(src, "pytest_gpu.log"), out("pytest_gpu.log"))
    stats = glob.glob(os.path.join(src, "prof_stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], out("bench_hor_m32_kernel_stats.csv"))

    rows = []
    for counter, sub in (("FETCH_SIZE", "prof_pmc_fetch"), ("WRITE_SIZE", "prof_pmc_write")):
        for kern, vals in sorted(pmc_rows(os.path.join(src, sub), counter).items()):
            if not kern.startswith(("void sg::", "sg::")):
                continue
            rows.append((kern.replace(",", ";"), counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
    with open(out("bench_hor_m32_pmc_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
        for r in rows:
            f.write("%s,%s,%d,%.3f,%.3f,%.3f\n" % r)

    # HBM bytes per launch of the bench kernels: 2 x FETCH_SIZE (gfx950 reports half the bytes of
    # a 16 B/lane stream; probe_read over exactly 1 GiB confirms it) + WRITE_SIZE
    def mean(kern_sub, counter):
        for r in rows:
            if kern_sub in r[0] and r[1] == counter:
                return r[3]
        return None

    traffic_path = os.path.join(here, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path))
    for kern, key in (("hor_scan", "hor_m32_sigma128_gib1"), ("packed_scan", "epsm_m32_sigma128_gib1")):
        fe, wr = mean(kern, "FETCH_SIZE"), mean(kern, "WRITE_SIZE")
        if fe is not None and wr is not None:
            traffic.setdefault(kern, {})[key] = int(round(2 * fe * 1024 + wr * 1024))
    probe = mean("probe_read", "FETCH_SIZE")
    traffic["_how"] = (
        "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 5 "
        "--warmup 2 --no-cpu`; bytes = 2*FETCH_SIZE_KiB*1024 (gfx950 reports half the bytes of a 16 B/lane "
        "coalesced stream, MI355X_MICROARCH.md §HBM) + WRITE_SIZE_KiB*1024; per launch; summaries in "
        f"profiles/{rnd}/ (latest: {prefix}_bench_hor_m32_pmc_summary.csv; the read probe with a known byte count "
        f"reads {probe:.0f} KiB in FETCH_SIZE for exactly 1 GiB, confirming the x2 correction for this access pattern)"
    )
    json.dump(traffic, open(traffic_path, "w"), indent=1)
    print("wrote", sorted(os.path.basename(p) for p in glob.glob(out("*"))))
    print(json.dumps({k: v for k, v in traffic.items() if k != "_how"}))


if __name__ == "__main__":
    main()
