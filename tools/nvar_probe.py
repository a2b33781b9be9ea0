import sys, numpy as np
sys.path.insert(0, '.')
import smart_amd
from smart_amd import Text
n = 1 << 30
text = Text.generate(0x5EED0001, 128, n)
for m in (8, 32):
    P = text.pattern(123456789, m)
    for algo in ("kmp", "so"):
        for sub in (n, n - 26 * 4096, n - 100000, n - (1 << 20), n - 64 * 4096, n - 65 * 4096, n - 122 * 64 * 4096, n):
            ts = []
            for r in range(12):
                c, pre, run = smart_amd.search(algo, P, text, off=0, n=sub)
                ts.append(run)
            ts = sorted(ts)[:6]
            print(f"{algo} m={m} n=2^30-{n-sub:<10d} run {np.mean(ts):.4f} ms  -> {sub/np.mean(ts)/1e6:.0f} GB/s count {c}")
