#!/usr/bin/env python3
"""Copy the judged summaries of one tools/gpu_round.sh session from gpurun_out/<tag>/ (scratch)
into profiles/<round>/<prefix>_* (tracked) and refresh profiles/pmc_traffic.json.

    python tools/collect_profiles.py r02a r02 a

Writes: <prefix>_bench_default.json, <prefix>_bench_hor_m32_kernel_stats.csv (rocprofv3 --stats),
<prefix>_bench_pmc_summary.csv (FETCH_SIZE / WRITE_SIZE per kernel, averaged over dispatches, KiB as
reported), <prefix>_pytest_gpu.log.  profiles/pmc_traffic.json records, besides the bytes per launch,
the sha256 of each kernel family's sources (smart_amd/sources.py) the passes were taken on and the commit: bench.py reports
roofline.traffic only for that kernel source.  Refuses a session that gpu_round.sh marked failed.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
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
    if os.path.exists(os.path.join(src, "SESSION_FAILED")):
        raise SystemExit("session %s failed (parity or bench): nothing collected" % tag)
    dst = os.path.join(here, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out = lambda name: os.path.join(dst, f"{prefix}_{name}")  # noqa: E731

    shutil.copy(os.path.join(src, "bench.json"), out("bench_default.json"))
    shutil.copy(os.path.join(src, "pytest_gpu.log"), out("pytest_gpu.log"))
    if os.path.exists(os.path.join(src, "bench_sweep.json")):  # the cells behind the line's min_frac / own_kernel_min
        shutil.copy(os.path.join(src, "bench_sweep.json"), out("bench_sweep.json"))
    stats = glob.glob(os.path.join(src, "prof_stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], out("bench_hor_m32_kernel_stats.csv"))

    rows = []
    for sub in sorted(glob.glob(os.path.join(src, "prof_pmc_*"))):
        if not os.path.isdir(sub):
            continue
        counter = "FETCH_SIZE" if "_fetch_" in os.path.basename(sub) else "WRITE_SIZE"
        algo = os.path.basename(sub).rsplit("_", 1)[1]  # hor, kmp, ... or kmp-sigma2, kmp-english (gpu_round.sh pmc)
        for kern, vals in sorted(pmc_rows(sub, counter).items()):
            if not kern.startswith(("void sg::", "sg::")):
                continue
            rows.append((algo, kern.replace(",", ";"), counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
    with open(out("bench_pmc_summary.csv"), "w") as f:
        f.write("bench_algo,kernel,counter,dispatches,mean_KiB,min_KiB,max_KiB\n")
        for r in rows:
            f.write("%s,%s,%s,%d,%.3f,%.3f,%.3f\n" % r)

    # HBM bytes per launch of the bench kernels: 2 x FETCH_SIZE (gfx950 reports half the bytes of
    # a 16 B/lane stream; probe_read over exactly 1 GiB confirms it) + WRITE_SIZE
    def mean(algo, kern_sub, counter):
        for r in rows:
            if r[0] == algo and kern_sub in r[1] and r[2] == counter:
                return r[4]
        return None

    def scan_kernel(algo):
        """the scan kernel that pass's bench ran most often: its name without template arguments"""
        best = None
        for r in rows:
            name = r[1].split("sg::", 1)[1].split("<")[0].split("(")[0]
            if r[0] == algo and r[2] == "FETCH_SIZE" and (name.endswith(("_scan", "_runs", "_scan_bp", "_gram"))) and (best is None or r[3] > best[1]):
                best = (name, r[3])
        return best[0] if best else None

    # the kernel sources the session ran: the per-family sha256 gpu_round.sh took on the box (smart_amd/sources.py);
    # a session without them is bound to the COMMITTED sources (the working tree may have moved on since the call started)
    sys.path.insert(0, here)
    from smart_amd import sources
    commit = subprocess.run(["git", "-C", here, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

    def committed(path):
        rel = os.path.relpath(path, here)
        return subprocess.run(["git", "-C", here, "show", "HEAD:" + rel], capture_output=True).stdout

    head_shas = sources.all_unit_shas(read=committed)
    sha_file = os.path.join(src, "kernel_sources.sha256.json")
    if os.path.exists(sha_file):
        shas = json.load(open(sha_file))
        moved = sorted(u for u in shas if shas[u] != head_shas.get(u))
        if moved:
            commit += " (session ran sources of %s that differ from this commit's)" % ", ".join(moved)
    else:
        shas = head_shas
    traffic = {"_source": {"commit": commit, "summary": f"profiles/{rnd}/{prefix}_bench_pmc_summary.csv"}}
    for algo in sorted({r[0] for r in rows}):
        kern = scan_kernel(algo)
        if kern is None:
            continue
        fe, wr = mean(algo, "sg::" + kern, "FETCH_SIZE"), mean(algo, "sg::" + kern, "WRITE_SIZE")
        name, _, variant = algo.partition("-")  # bench.py's workload key: <algo>_m<m>_sigma<s>_gib<g> / <algo>_m<m>_english_gib<g>
        key = "%s_m32_%s_gib1" % (name, variant if variant else "sigma128")
        if fe is not None and wr is not None:
            traffic.setdefault(kern, {"_sha256": shas[sources.KERNEL_UNIT[kern]]})[key] = int(round(2 * fe * 1024 + wr * 1024))
    probe = mean("hor", "probe_read", "FETCH_SIZE")
    traffic["_how"] = (
        "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --algo <a> --steps 5 "
        "--warmup 2 --no-cpu --no-sweep`; bytes = 2*FETCH_SIZE_KiB*1024 (gfx950 reports half the bytes of a 16 B/lane "
        "coalesced stream, MI355X_MICROARCH.md §HBM) + WRITE_SIZE_KiB*1024; per launch; "
        + ("the read probe with a known byte count reads %.0f KiB in FETCH_SIZE for exactly 1 GiB, confirming the x2 "
           "correction for this access pattern" % probe if probe is not None else "no read-probe row in this session"))
    json.dump(traffic, open(os.path.join(here, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.path.basename(p) for p in glob.glob(out("*"))))
    print(json.dumps({k: v for k, v in traffic.items() if k != "_how"}))


if __name__ == "__main__":
    main()
