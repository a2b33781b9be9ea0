#!/usr/bin/env python3
"""Does the NUMBER of occurrences cost the packed matcher time?  The same text, patterns of one length that occur
often / never:  python tools/hits_probe.py [gib] [algo]   (kernel ms by the plan's HIP events, median of 15)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smart_amd
from smart_amd.engine import Text

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
algo = sys.argv[2] if len(sys.argv) > 2 else "epsm"
n = int(gib * (1 << 30))
text = Text.generate(0x5EED0001, 128, n)
for m in (2, 3, 4, 8):
    present = text.read(12345, m)
    absent = np.full(m, 200, dtype=np.uint8)   # no byte of rand128 is 200
    half = present.copy()
    half[-1] = 200                              # its first bytes occur, the pattern does not
    for name, P in (("present", present), ("absent", absent), ("prefix", half)):
        ts, c = [], 0
        for _ in range(15):
            c, _, run = smart_amd.search(algo, P, text)
            ts.append(run)
        ts.sort()
        print("%s m=%d %-8s count %-10d median %.4f ms  min %.4f  = %.1f %% of 8 TB/s" % (algo, m, name, c, ts[7], ts[0], n / ts[7] / 1e6 / 8000 * 100))
