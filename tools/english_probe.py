#!/usr/bin/env python3
"""Where does the time go on English for long patterns?  Times each algorithm on a pattern taken from the
text and on copies with one byte changed (first / middle / last), which removes the occurrences but keeps
most of the partial matches:  python tools/english_probe.py [--m 64] [--algos bm,bndm,hor]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smart_amd  # noqa: E402
from smart_amd import Plan, Text  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--gib", type=float, default=1.0)
ap.add_argument("--algos", default="hor,bm,bndm,sbndm,qs,bndml,epsm")
ap.add_argument("--at", type=int, default=100003)
args = ap.parse_args()
n = int(args.gib * (1 << 30))
unit = np.fromfile(os.path.join(ROOT, "tests", "golden", "english_excerpt.txt"), dtype=np.uint8)[:262139]
text = Text.upload_tiled(unit, n)
m = args.m
base = unit[args.at:args.at + m].copy()
variants = {"as is": base}
for name, i in (("first byte changed", 0), ("middle byte changed", m // 2), ("last byte changed", m - 1)):
    v = base.copy()
    v[i] = ord("#")
    variants[name] = v
print("pattern: %r" % bytes(base[:64]))
for algo in args.algos.split(","):
    for name, P in variants.items():
        pl = Plan(algo, P)
        pl.launch(text, slot=1)
        pl.result(1)
        ts = []
        for _ in range(3):
            pl.launch(text, slot=0, timed=True)
            c, t = pl.result(0)
            ts.append(t)
        t = sorted(ts)[1]
        print("%-6s %-20s %-12s count %-7d %7.4f ms  %5.1f%% of 8 TB/s" % (algo, name, pl.kernel_name, c, t, n / (t * 1e-3) / 8e10), flush=True)
        pl.free()
