#!/usr/bin/env python3
"""Rewrite every table of DESIGN.md §8 from one session's files in profiles/<round>/ and print the numbers its prose quotes:
    python tools/design_section8.py a r03
(the prose between the tables is kept as it stands; tables are matched in the order they appear)."""
import csv
import json
import re
import subprocess
import sys

prefix = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
root = "profiles/%s/%s_" % (rnd, prefix)


def tab(name):
    out = subprocess.run([sys.executable, "tools/tables.py", root + name + ".log"], capture_output=True, text=True).stdout
    return "\n".join(line for line in out.splitlines() if line.startswith("|"))


s = open("DESIGN.md").read()
a = s.index("## 8. Round-%d results" % int(rnd[1:]))
head, body = s[:a], s[a:]
# tables, in the order they appear
order = ["sweep_rand128_full", "sweep_rand128_own", "sweep_rand4", "sweep_rand2", "sweep_rand4_own", "sweep_rand2_own",
         "sweep_english_4gib", "sweep_english_4gib_own", "sweep_cfg5_rand2_4gib", "sweep_cfg5_rand32_4gib", "sweep_cfg5_rand256_4gib",
         "sweep_f3_rand128", "sweep_f3_rand4", "sweep_f3_english_4gib"]
blocks = re.findall(r"(?:^\|.*\n)+", body, flags=re.M)
assert len(blocks) == len(order), (len(blocks), len(order))
for old, name in zip(blocks, order):
    body = body.replace(old, tab(name) + "\n", 1)
body = re.sub(r"profiles/%s/[a-z]_sweep_" % rnd, "profiles/%s/%s_sweep_" % (rnd, prefix), body)
body = re.sub(r"`[a-z]_pytest_gpu.log`", "`%s_pytest_gpu.log`" % prefix, body)
body = re.sub(r"`[a-z]_bench_default.json`", "`%s_bench_default.json`" % prefix, body)
body = re.sub(r"`[a-z]_bench_pmc_summary.csv`", "`%s_bench_pmc_summary.csv`" % prefix, body)
open("DESIGN.md", "w").write(head + body)

d = json.load(open(root + "bench_default.json"))
r = d["roofline"]
pm = {}
for row in csv.DictReader(open(root + "bench_pmc_summary.csv")):
    pm[(row["bench_algo"], row["kernel"].split("sg::")[1].split("<")[0].split("(")[0], row["counter"])] = float(row["mean_KiB"])


def ratio(algo, k):
    return (2 * pm[(algo, k, "FETCH_SIZE")] * 1024 + pm[(algo, k, "WRITE_SIZE")] * 1024) / 2**30


hs = [x for x in csv.DictReader(open(root + "bench_hor_m32_kernel_stats.csv")) if "hor_scan" in x["Name"]][0]
print("headline: %.2f TB/s = %.1f %% of 8 TB/s = %.0f %% of stream; kernel %.4f ms (events), %.4f ms (rocprofv3 --stats, %s launches); "
      "traffic %s B = %.3fx" % (d["value"] / 1000, r["frac"] * 100, r["frac_of_measured_stream_read"] * 100, r["kernel_ms"],
                               float(hs["AverageNs"]) / 1e6, hs["Calls"], "{:,}".format(r["traffic"]) if r["traffic"] else None,
                               (r["traffic"] or 0) / 2**30))
print("pmc ratios:", {k: round(ratio(*k[:2]), 3) for k in pm if k[2] == "FETCH_SIZE" and k[1].endswith(("_runs", "_scan"))})
print("cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"])
print("min_frac:", d["min_frac"], "cells", d.get("sweep_cells"))
print("own_kernel_min:", d.get("own_kernel_min"))
