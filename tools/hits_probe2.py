import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import smart_amd
from smart_amd import engine
from smart_amd.engine import Text
n = 4 << 30
text = Text.generate(0x5EED0001, 128, n)
def med(algo, P):
    ts = []
    for _ in range(11):
        c, _, run = smart_amd.search(algo, P, text)
        ts.append(run)
    ts.sort()
    return c, ts[5]
present = text.read(12345, 2); absent = np.full(2, 200, dtype=np.uint8)
for wgs in (16, 8, 4, 32):
    engine.tune(4, wgs)
    c1, t1 = med("epsm", present); c0, t0 = med("epsm", absent)
    print("epsm wgs/CU %2d: present %.4f absent %.4f  diff %.1f us (count %d)" % (wgs, t1, t0, (t1 - t0) * 1000, c1))
engine.tune(4, 0)
for algo in ("so", "kmp", "hor", "bndm"):
    for m in (2, 12):
        present = text.read(12345, m); absent = np.full(m, 200, dtype=np.uint8)
        c1, t1 = med(algo, present); c0, t0 = med(algo, absent)
        print("%s m=%d [%s]: present %.4f absent %.4f diff %.1f us (count %d)" % (algo, m, smart_amd.kernel_for(algo, present), t1, t0, (t1 - t0) * 1000, c1))
