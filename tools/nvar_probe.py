"""Same text, same pattern, different lengths of the searched range: what a partly filled last group of runs or a
short last run costs the runs kernels (kmp_runs, so_runs).  python tools/nvar_probe.py [m ...]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import smart_amd
from smart_amd import Text
n = (1 << 30) + 8192
text = Text.generate(0x5EED0001, 128, n)
ms = [int(x) for x in sys.argv[1:]] or [8, 32]
for m in ms:
    P = text.pattern(123456789, m)
    full = (1 << 30) + m - 1  # every run of 4096 bytes has all its start positions
    for algo in ("kmp", "so"):
        for sub, what in ((full, "all runs full"), (full - 4095, "last run: 1 start"), (full - 2048, "last run: half"),
                          (full - 26 * 4096, "last group: 38 runs"), (full - 64 * 4096, "one group fewer"),
                          (full - 122 * 64 * 4096, "122 groups fewer"), (full, "all runs full")):
            ts = []
            for r in range(12):
                c, pre, run = smart_amd.search(algo, P, text, off=0, n=sub)
                ts.append(run)
            ts = sorted(ts)[:6]
            print(f"{algo} m={m:<5d} {what:22s} run {np.mean(ts):.4f} ms  -> {sub/np.mean(ts)/1e6:.0f} GB/s count {c}")
