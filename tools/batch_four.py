#!/usr/bin/env python3
"""Pattern sets on a small four-symbol text (SMART's 1 MiB stock size): per-pattern wall time of one batch call, with the
runs kernels' four-bytes-per-step tables (built by every workgroup when it starts) and without (tune(3,5), (6,5))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smart_amd
from smart_amd import Text, engine

rng = np.random.default_rng(5)
for sigma in (4, 2):
    T = rng.integers(0, sigma, 1 << 20, dtype=np.uint8)
    text = Text.upload(T)
    for algo in ("kmp", "so"):
        for m in (8, 32):
            pats = [T[k:k + m].copy() for k in rng.integers(0, len(T) - m, 500)]
            for key, val, name in ((3, 0, "four"), (3, 5, "byte")) if algo == "kmp" else ((6, 0, "four"), (6, 5, "byte")):
                engine.tune(key, val)
                best = 1e9
                for rep in range(5):
                    t0 = time.perf_counter()
                    smart_amd.search_batch(algo, pats, text, per_pattern_times=False)
                    best = min(best, time.perf_counter() - t0)
                engine.tune(key, 0)
                print("sigma %d %-3s m=%-3d %-4s %.2f us per pattern (%s)" % (sigma, algo, m, name, best / len(pats) * 1e6, smart_amd.kernel_for(algo, pats[0])))
    text.free()
