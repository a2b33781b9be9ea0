#!/usr/bin/env python3
"""Differential fuzz on a GPU box: random texts, patterns with borders / periods / planted copies, random sub-ranges —
every algorithm through the C ABI against the oracle's brute force.  python tools/fuzz_gpu.py [seconds] [seed]
(not a test: run by hand after changes to the kernels; the committed parity suite is tests/test_parity_gpu.py)"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import smart_amd
from smart_amd import Text, engine
from oracle import pyoracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = checks = 0
bad = []
while time.time() < t_end and not bad:
    sigma = int(rng.choice([2, 2, 3, 4, 4, 8, 32, 128, 256]))
    n = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 300_000), rng.integers(300_000, 3_000_000)]))
    T = rng.integers(0, sigma, n, dtype=np.uint8)
    kind = rng.integers(0, 5)
    mmax = int(rng.choice([8, 40, 70, 300, 700, 4200]))
    m = int(min(max(1, rng.integers(1, mmax + 1)), max(1, n)))
    if kind == 0 and n > m:      # a piece of the text
        k = int(rng.integers(0, n - m + 1))
        P = T[k:k + m].copy()
    elif kind == 1:              # periodic: a short unit repeated
        u = rng.integers(0, sigma, int(rng.integers(1, 6)), dtype=np.uint8)
        P = np.resize(u, m)
    elif kind == 2:              # bordered: u v u
        lu = max(1, m // int(rng.integers(2, 5)))
        u = rng.integers(0, sigma, lu, dtype=np.uint8)
        P = np.concatenate([u, rng.integers(0, sigma, max(0, m - 2 * lu), dtype=np.uint8), u])[:m]
        m = len(P)
    elif kind == 3:              # almost periodic: one byte changed
        u = rng.integers(0, sigma, int(rng.integers(1, 9)), dtype=np.uint8)
        P = np.resize(u, m).copy()
        P[int(rng.integers(0, m))] = rng.integers(0, sigma)
    else:
        P = rng.integers(0, sigma, m, dtype=np.uint8)
    P = np.ascontiguousarray(P, dtype=np.uint8)
    m = len(P)
    if n >= m and rng.integers(0, 2):   # plant copies, some overlapping, one at the very end
        for _ in range(int(rng.integers(1, 40))):
            k = int(rng.integers(0, n - m + 1))
            T[k:k + m] = P
        if rng.integers(0, 2):
            T[n - m:] = P
    if rng.integers(0, 3) == 0:  # the symbols renamed to arbitrary byte values (texts of two to four values that are not 0..3)
        perm = rng.permutation(256).astype(np.uint8)
        T = perm[T]
        P = np.ascontiguousarray(perm[P])
    text = Text.upload(T)
    ranges = [(0, n)]
    for _ in range(2):
        off = int(rng.integers(0, n))
        ranges.append((off, int(rng.integers(0, n - off + 1))))
    own = bool(rng.integers(0, 4) == 0)
    if own:
        engine.tune(0, 1)
    try:
        for off, nn in ranges:
            want = pyoracle.search("bf", P, T[off:off + nn]) if nn >= m else 0
            for a in smart_amd.ALGOS:
                if m < smart_amd.MIN_M.get(a, 1):
                    continue
                got = smart_amd.search(a, P, text, off=off, n=nn)[0]
                checks += 1
                if got != want:
                    bad.append((a, sigma, n, m, kind, off, nn, own, got, want))
    finally:
        if own:
            engine.tune(0, 0)
    text.free()
    cases += 1
print("fuzz: %d cases, %d checks, seed %d: %s" % (cases, checks, seed, "ALL EQUAL" if not bad else "MISMATCH %s" % bad[:5]))
sys.exit(1 if bad else 0)
