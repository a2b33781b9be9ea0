"""Differential fuzz under -m gpu: the random sweep src/test.c:228-250 leaves commented out, closed here.

Seeded and bounded: random texts (sigma 2 .. 256, 1 byte .. 2 MB), pattern sets that are pieces of the text,
periodic, bordered (u v u), almost periodic or random, planted copies (overlapping, one at the very end), random
sub-ranges (what a shard sees), symbols renamed to arbitrary byte values — every algorithm through the C ABI, as a pattern
set in one call (one grid, or every pattern on its own: smartgpu_search_batch64_each) and call by call,
on the plan's kernel and on its own (smartgpu_tune(0,1)), against the oracle's brute force (bf.c:25-39).
400,000 comparisons; tools/fuzz_gpu.py is the open-ended form of the same generator.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import smart_amd  # noqa: E402
from smart_amd import Text, engine  # noqa: E402

CHECKS = 400_000   # ~25 s on an MI355X box (100,000 took 5.5 s)
SEED = 20261004


def draw_pattern(rng, T, sigma, m, kind):
    n = len(T)
    if kind == 0 and n > m:      # a piece of the text
        k = int(rng.integers(0, n - m + 1))
        return T[k:k + m].copy()
    if kind == 1:                # periodic: a short unit repeated
        return np.resize(rng.integers(0, sigma, int(rng.integers(1, 6)), dtype=np.uint8), m)
    if kind == 2:                # bordered: u v u
        lu = max(1, m // int(rng.integers(2, 5)))
        u = rng.integers(0, sigma, lu, dtype=np.uint8)
        return np.resize(np.concatenate([u, rng.integers(0, sigma, max(0, m - 2 * lu), dtype=np.uint8), u]), m)
    if kind == 3:                # almost periodic: one byte changed
        P = np.resize(rng.integers(0, sigma, int(rng.integers(1, 9)), dtype=np.uint8), m).copy()
        P[int(rng.integers(0, m))] = rng.integers(0, sigma)
        return P
    return rng.integers(0, sigma, m, dtype=np.uint8)


def test_differential_fuzz_every_algorithm_both_routings(oracle):
    assert smart_amd.device_count() > 0
    rng = np.random.default_rng(SEED)
    t0 = time.time()
    checks = cases = 0
    by_kernel = {}
    while checks < CHECKS:
        sigma = int(rng.choice([2, 2, 3, 4, 4, 8, 32, 128, 256]))
        n = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 100_000), rng.integers(3000, 100_000), rng.integers(100_000, 2_000_000)]))
        T = rng.integers(0, sigma, n, dtype=np.uint8)
        m = int(min(max(1, rng.integers(1, int(rng.choice([8, 40, 70, 300, 700, 4200])) + 1)), n))
        pats = [np.ascontiguousarray(draw_pattern(rng, T, sigma, m, int(rng.integers(0, 5))), dtype=np.uint8) for _ in range(6)]
        for P in pats[:3]:       # plant copies of half of the set, some overlapping, one at the very end
            for _ in range(int(rng.integers(1, 20))):
                k = int(rng.integers(0, n - m + 1))
                T[k:k + m] = P
        if rng.integers(0, 2):
            T[n - m:] = pats[0]
        if cases % 3 == 2:       # every third case: the symbols renamed to arbitrary byte values (a text of two, three or four
            perm = rng.permutation(256).astype(np.uint8)  # values that are not 0..3: the codes of the four-byte and gram tables)
            T = perm[T]
            pats = [np.ascontiguousarray(perm[P]) for P in pats]
        text = Text.upload(T)
        ranges = [(0, n)]
        off = int(rng.integers(0, n))
        ranges.append((off, int(rng.integers(0, n - off + 1))))
        own = cases % 2 == 1     # every second case: each algorithm on its own kernel
        if own:
            engine.tune(0, 1)
        try:
            for off, nn in ranges:
                want = [oracle.search("bf", P, T[off:off + nn]) if nn >= m else 0 for P in pats]
                for a in smart_amd.ALGOS:
                    if m < smart_amd.MIN_M.get(a, 1):
                        continue
                    k = engine.kernel_for(a, pats[0])
                    by_kernel[k] = by_kernel.get(k, 0) + len(pats)
                    if nn >= m:  # the pattern set in one call (smart.c:312-345 as one launch group)
                        got = smart_amd.search_batch(a, pats, text, off=off, n=nn, per_pattern_times=cases % 4 == 0, each=cases % 4 == 0)[0].tolist()
                        assert got == want, ("batch", a, sigma, n, m, off, nn, own, got, want, cases)
                        checks += len(pats)
                    j = int(rng.integers(0, len(pats)))  # and one of them call by call
                    got1 = smart_amd.search(a, pats[j], text, off=off, n=nn)[0]
                    assert got1 == want[j], ("call", a, sigma, n, m, off, nn, own, j, got1, want[j], cases)
                    checks += 1
        finally:
            if own:
                engine.tune(0, 0)
        text.free()
        cases += 1
        assert time.time() - t0 < 600, "fuzz too slow: %d checks in %d cases" % (checks, cases)
    print("fuzz: %d cases, %d checks in %.1f s; by kernel %s" % (cases, checks, time.time() - t0, by_kernel))
    assert checks >= CHECKS and {"hor_scan", "bm_scan", "bndm_scan", "kmp_runs", "so_runs", "packed_scan"} <= set(by_kernel)
