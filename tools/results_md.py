#!/usr/bin/env python3
"""Write profiles/<round>/RESULTS.md — the result tables of one session — from its files in profiles/<round>/<letter>_*:

    python tools/results_md.py p r04 [r]      (r: the session letter of the bench line, if it was re-taken)

(the sweeps of tools/sweep_all.sh and tools/own_sweep.sh rendered by tools/tables.py, the headline from the bench line,
the PMC ratios from the session's summary).  DESIGN.md points here; nothing in this file is written by hand."""
import csv
import json
import os
import subprocess
import sys

prefix, rnd = sys.argv[1], sys.argv[2]
head_prefix = sys.argv[3] if len(sys.argv) > 3 else prefix  # the bench line re-taken after the session's PMC passes were collected
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.join(ROOT, "profiles", rnd, prefix + "_")
head_root = os.path.join(ROOT, "profiles", rnd, head_prefix + "_")


def tab(name):
    path = root + name + ".log"
    if not os.path.exists(path):
        return None
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tables.py"), path], capture_output=True, text=True).stdout
    return "\n".join(line for line in out.splitlines() if line.startswith("|"))


SECTIONS = [
    ("rand128, 1 GiB (BASELINE config 2; north-star bar: >= 60 % for m in {4..256}) — the plan's kernel choice", "sweep_rand128_full"),
    ("rand128, every algorithm on its OWN kernel (smartgpu_tune(0,1))", "sweep_rand128_own"),
    ("sigma = 4, 1 GiB (config 3) — the plan's choice", "sweep_rand4"),
    ("sigma = 4 — own kernels", "sweep_rand4_own"),
    ("sigma = 2, 1 GiB (config 3) — the plan's choice", "sweep_rand2"),
    ("sigma = 2 — own kernels", "sweep_rand2_own"),
    ("English (bible.txt || world192.txt) tiled to 4 GiB (config 4) — the plan's choice", "sweep_english_4gib"),
    ("English, 4 GiB — own kernels", "sweep_english_4gib_own"),
    ("config 5: sigma = 2, one 4 GiB shard", "sweep_cfg5_rand2_4gib"),
    ("config 5: sigma = 32, one 4 GiB shard", "sweep_cfg5_rand32_4gib"),
    ("config 5: sigma = 256, one 4 GiB shard", "sweep_cfg5_rand256_4gib"),
    ("the adjacent algorithms (f3) on rand128", "sweep_f3_rand128"),
    ("f3 on sigma = 4", "sweep_f3_rand4"),
    ("f3 on English, 4 GiB", "sweep_f3_english_4gib"),
]

out = ["# Results of session `%s` (%s, 1x MI355X; kernel time by HIP events; %% of the 8 TB/s HBM peak)" % (prefix, rnd), "",
       "Written by `tools/results_md.py %s %s` from `profiles/%s/%s_*` — the raw logs of `tools/gpu_round.sh`, `tools/sweep_all.sh`," % (prefix, rnd, rnd, prefix),
       "`tools/own_sweep.sh` (every cell's counts cross-checked between the kernels).  A mark on a cell says that its plans did not run",
       "on the algorithm's own kernel: **s** = `so_runs`, **p** = `packed_scan`, **~** = only some of the cell's three patterns.", ""]

bench = head_root + "bench_default.json"
if os.path.exists(bench):
    d = json.loads(open(bench).read())
    r = d["roofline"]
    out += ["## Headline (`bench.py`, %s; `%s_bench_default.json`)" % (d["config"]["workload"], head_prefix), "",
            "* **%.2f TB/s = %.1f %% of 8 TB/s** (= %.0f %% of the measured streaming read, %.0f GB/s); kernel %.4f ms by HIP events over %d launches."
            % (d["value"] / 1000, r["frac"] * 100, r["frac_of_measured_stream_read"] * 100, r["measured_stream_read_GBps"], r["kernel_ms"], d["steps"]),
            "* HBM traffic by PMC: %s" % ("%s B = %.3fx the algorithmic bytes" % ("{:,}".format(r["traffic"]), r["traffic"] / r["bytes_per_launch"]) if r.get("traffic") else "null (%s)" % r.get("traffic_source")),
            "* CPU baseline: %s" % (json.dumps(d["cpu_baseline"])[:400] if d.get("cpu_baseline") else "not run"),
            "* `min_frac` (plan's choice): `%s`" % json.dumps(d.get("min_frac")),
            "* `own_kernel_min`: `%s`" % json.dumps(d.get("own_kernel_min")),
            "* `worst_cells`: `%s`" % json.dumps(d.get("worst_cells")), ""]
    stats = head_root + "bench_hor_m32_kernel_stats.csv"
    if os.path.exists(stats):
        rows = [x for x in csv.DictReader(open(stats)) if r["kernel"] in x["Name"]]
        if rows:
            out += ["* rocprofv3 `--kernel-trace --stats` of the same command: %s average %.4f ms over %s launches." % (r["kernel"], float(rows[0]["AverageNs"]) / 1e6, rows[0]["Calls"]), ""]
pmc = root + "bench_pmc_summary.csv"
if os.path.exists(pmc):
    pm = {}
    for row in csv.DictReader(open(pmc)):
        name = row["kernel"].split("sg::")[1].split("<")[0].split("(")[0]
        pm[(row["bench_algo"], name, row["counter"])] = float(row["mean_KiB"])
    ratios = {}
    for (algo, k, c) in pm:
        if c == "FETCH_SIZE" and k.endswith(("_runs", "_scan")) and (algo, k, "WRITE_SIZE") in pm:
            ratios["%s / %s" % (algo, k)] = round((2 * pm[(algo, k, "FETCH_SIZE")] + pm[(algo, k, "WRITE_SIZE")]) * 1024 / 2**30, 3)
    out += ["## HBM traffic per launch / algorithmic bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE, 1 GiB texts)", "", "`%s`" % json.dumps(ratios), ""]

for title, name in SECTIONS:
    t = tab(name)
    if t:
        out += ["## " + title, "", "`%s_%s.log`" % (prefix, name), "", t, ""]
for extra in ("own_english", "own_sigma4", "own_sigma2", "own_sigma128", "own_kmp_english", "own_kmp_sigma2"):
    t = tab(extra)
    if t:
        out += ["## own kernels at 1 GiB: " + extra, "", t, ""]
open(os.path.join(ROOT, "profiles", rnd, "RESULTS.md"), "w").write("\n".join(out) + "\n")
print("wrote profiles/%s/RESULTS.md (%d lines)" % (rnd, len(out)))
