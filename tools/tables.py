#!/usr/bin/env python3
"""Render tools/sweep.py logs as the markdown tables of DESIGN.md §8: python tools/tables.py <log> [...]"""
import re
import sys

ROW = re.compile(r"^(\w+)\s+m=(\d+)\s+sigma=(\d+)\s+([\d.]+) ms.*?([\d.]+) GB/s\s+([\d.]+)% of")
ORDER = ["hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita", "hash3", "hash5", "hash8", "sbndm", "kr", "bndml"]


def table(path):
    cells, ms = {}, []
    for line in open(path):
        mt = ROW.match(line)
        if not mt:
            continue
        algo, m, pct = mt.group(1), int(mt.group(2)), float(mt.group(6))
        cells[(algo, m)] = pct
        if m not in ms:
            ms.append(m)
    algos = [a for a in ORDER if any((a, m) in cells for m in ms)]
    out = ["| algo | " + " | ".join("m=%d" % m if i == 0 else str(m) for i, m in enumerate(ms)) + " |",
           "|---" * (len(ms) + 1) + "|"]
    for a in algos:
        out.append("| %s | " % a.upper() + " | ".join("%.0f%%" % cells[(a, m)] if (a, m) in cells else "" for m in ms) + " |")
    return "\n".join(out)


for p in sys.argv[1:]:
    print(p)
    print(table(p))
    print()
