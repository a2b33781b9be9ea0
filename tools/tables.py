#!/usr/bin/env python3
"""Render tools/sweep.py logs as the markdown tables of DESIGN.md §8: python tools/tables.py <log> [...]
A cell whose plans did not run on the algorithm's own kernel (api.cpp build_blob rerouted the pattern) carries a
mark: p = packed_scan, s = so_runs; mixed = some of the cell's patterns each."""
import re
import sys

ROW = re.compile(r"^(\w+)\s+m=(\d+)\s+sigma=(\d+)\s+([\d.]+) ms.*?([\d.]+) GB/s\s+([\d.]+)% of.*?(?:\[([\w+]+)\])?\s*$")
ORDER = ["hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita", "hash3", "hash5", "hash8", "sbndm", "kr", "bndml"]
OWN = {"hor": "hor_scan", "bm": "bm_scan", "kmp": "kmp_runs", "so": "so_runs", "bndm": "bndm_scan", "epsm": "packed_scan",
       "sa": "so_runs", "qs": "hor_scan", "tunedbm": "hor_scan", "raita": "hor_scan", "hash3": "hor_scan", "hash5": "hor_scan",
       "hash8": "hor_scan", "sbndm": "sbndm_scan", "kr": "hor_scan_bp", "bndml": "bndml_scan"}
MARK = {"packed_scan": "p", "so_runs": "s"}


def mark(algo, m, kernels):
    if not kernels:
        return ""
    ks = kernels.split("+")
    own = "bndm_scan" if algo == "bndml" and m <= 32 else OWN[algo]
    other = [k for k in ks if k != own]
    if not other:
        return ""
    return " " + "".join(sorted({MARK.get(k, "?") for k in other})) + ("~" if len(ks) > 1 and own in ks else "")


def table(path):
    cells, ms = {}, []
    for line in open(path):
        mt = ROW.match(line.rstrip("\n"))
        if not mt:
            continue
        algo, m, pct = mt.group(1), int(mt.group(2)), float(mt.group(6))
        cells[(algo, m)] = "%.0f%%%s" % (pct, mark(algo, m, mt.group(7)))
        if m not in ms:
            ms.append(m)
    algos = [a for a in ORDER if any((a, m) in cells for m in ms)]
    out = ["| algo | " + " | ".join("m=%d" % m if i == 0 else str(m) for i, m in enumerate(ms)) + " |",
           "|---" * (len(ms) + 1) + "|"]
    for a in algos:
        out.append("| %s | " % a.upper() + " | ".join(cells.get((a, m), "") for m in ms) + " |")
    return "\n".join(out)


for p in sys.argv[1:]:
    print(p)
    print(table(p))
    print("(p = the cell's plans ran on packed_scan, s = on so_runs, ~ = only some of its patterns)")
    print()
