"""Wall time of a synchronous smartgpu_search64 call (plan build + launch + count on the host) on SMART's stock 1 MiB text:
python tools/call_probe.py   (SMARTGPU_LIB selects the library)"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import smart_amd
from smart_amd import Text
n = 1 << 20
text = Text.generate(0x5EED0001, 128, n)
for algo in ("hor", "bm", "kmp", "so", "epsm"):
    for m in (8, 32, 256):
        pats = [text.pattern((1000 + 3301 * j) % (n - m), m) for j in range(300)]
        for p in pats[:20]:
            smart_amd.search(algo, p, text)
        t0 = time.perf_counter()
        tot = 0
        for p in pats:
            tot += smart_amd.search(algo, p, text)[0]
        dt = (time.perf_counter() - t0) / len(pats)
        print(f"{algo:5s} m={m:<4d} {dt*1e6:7.1f} us per call (counts {tot})")
