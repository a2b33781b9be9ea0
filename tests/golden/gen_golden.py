#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REAL reference
algorithms (oracle/_ref/lib<algo>.so, built by `make -C oracle ref` from
/root/reference/src/algos/<algo>.c).

Runs only where /root/reference exists (the build container).  The outputs are
data: inputs are described by generator parameters (seed, sigma, n, offset) or
short literals, expected outputs are the occurrence counts the reference
returned.  No reference source text is stored.

    python tests/golden/gen_golden.py

Files written:
  survey_vectors.json   SURVEY.md §8c starter table on textgen corpora (glibc rand stream)
  testc_cases.json      src/test.c:252-382 deterministic cases (1-11, 16-20)
  fuzz_vectors.json     seeded random cases over sigma/n/m incl. m>32, m>64, periodic texts
  deviations.json       documented reference deviations (EPSM tail miss, SO/BNDM straddle)
  english_vectors.json  counts on english_excerpt.txt (first 256 KiB of englishTexts/bible.txt)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = {a: po.RefAlgo(a) for a in po.ALGOS}


def ref_counts(P, T):
    """Counts of all seven reference algorithms.  EPSM is skipped for n<16
    (epsm.c:113-114 wraps an unsigned index and segfaults)."""
    out = {}
    for a, r in REF.items():
        if a == "epsm" and len(T) < 16:
            out[a] = None
            continue
        out[a] = r.search(P, T)
    return out


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", name)


def survey_vectors():
    rows = []
    n = 1048576  # default -tsize, smart.c:416
    for sigma, ms in ((128, (1, 2, 3, 4, 8, 16, 32, 64, 256, 4096)),
                      (2, (1, 2, 3, 4, 8, 16, 32, 64, 256, 4096)),
                      (4, (2, 4, 8, 16, 32))):
        T = po.textgen(sigma, n)
        for m in ms:
            for k in (0, 12345, 524288, n - m - 1):
                P = T[k:k + m]
                c = ref_counts(P, T)
                assert len(set(c.values())) == 1, (sigma, m, k, c)
                rows.append({"sigma": sigma, "n": n, "m": m, "k": k, "count": c["bf"]})
    dump("survey_vectors.json", {"text": "oracle_textgen(sigma)[:n] == src/textgen.c stream", "rows": rows})


def testc_cases():
    A = lambda s: s  # noqa: E731
    cases = [
        (1, "a", "aaaaaaaaaa"), (2, "aa", "aaaaaaaaaa"), (3, "aaaaaaaaaa", "aaaaaaaaaa"),
        (4, "b", "aaaaaaaaaa"), (5, "ab", "ababababab"), (6, "a", "ababababab"),
        (7, "aba", "ababababab"), (8, "abc", "ababababab"), (9, "ba", "ababababab"),
        (10, "babbbbb", "ababababab"), (11, "bcdefg", "bcdefghilm"),
        (16, "a" * 40, "a" * 64), (17, "ab" * 20, "ab" * 32),
        (18, "ab" * 19 + "ac", "ab" * 32),
        (19, "babbbbb", "abababbbbb"), (20, "bababb", "abababbbbb"),
    ]
    rows = []
    for no, p, t in cases:
        P = np.frombuffer(A(p).encode(), dtype=np.uint8)
        T = np.frombuffer(A(t).encode(), dtype=np.uint8)
        c = ref_counts(P, T)
        vals = {v for v in c.values() if v is not None}
        assert len(vals) == 1, (no, c)
        rows.append({"case": no, "P": p, "T": t, "count": c["bf"], "epsm_ref_ran": c["epsm"] is not None})
    dump("testc_cases.json", {"source": "src/test.c:252-382 (cases 12-15 use srand(time) and are not reproducible)",
                              "rows": rows})


def make_case(rng, kind):
    sigma = int(rng.choice([2, 3, 4, 16, 64, 128, 250, 256]))
    n = int(rng.integers(16, 6000))
    seed = int(rng.integers(0, 2**62))
    T = po.gen_text(seed, sigma, 0, n)
    m = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 96, 128, 200, 256, 300]))
    m = min(m, n)
    desc = {"seed": seed, "sigma": sigma, "n": n, "m": m, "kind": kind}
    if kind == "planted":
        k = int(rng.integers(0, n - m + 1))
        P = T[k:k + m].copy()
        desc["k"] = k
    elif kind == "mutated":
        k = int(rng.integers(0, n - m + 1))
        P = T[k:k + m].copy()
        pos = int(rng.integers(0, m))
        P[pos] = (int(P[pos]) + 1) % sigma
        desc["k"] = k
        desc["mut"] = pos
    elif kind == "periodic":
        # text = (unit)^* with a short unit -> many overlapping occurrences
        u = int(rng.integers(1, 5))
        unit = po.gen_text(seed, sigma, 0, u)
        T = np.resize(unit, n)
        k = int(rng.integers(0, min(n - m + 1, 4 * u)))
        P = T[k:k + m].copy()
        desc["unit"] = u
        desc["k"] = k
    elif kind == "tail":
        # pattern planted at the very end / very start (boundary coverage)
        k = n - m if rng.integers(0, 2) else 0
        P = T[k:k + m].copy()
        desc["k"] = k
    else:
        raise ValueError(kind)
    return desc, P, T


def fuzz_vectors(count=1600):
    rng = np.random.default_rng(20260310)
    rows = []
    kinds = ["planted"] * 5 + ["mutated"] * 2 + ["periodic"] * 2 + ["tail"] * 2
    while len(rows) < count:
        kind = kinds[len(rows) % len(kinds)]
        desc, P, T = make_case(rng, kind)
        c = ref_counts(P, T)
        truth = c["bf"]
        dev = {a: v for a, v in c.items() if v is not None and v != truth}
        # the only deviation the reference is allowed to show here is the EPSM
        # tail miss (epsm.c:330): m%8==0, m>=16, occurrence at s=n-m
        for a, v in dev.items():
            m, n = desc["m"], desc["n"]
            assert a == "epsm" and m >= 16 and m % 8 == 0 and v == truth - 1 and \
                np.array_equal(T[n - m:], P), (desc, c)
        desc["count"] = truth
        if dev:
            desc["ref_deviation"] = dev
        rows.append(desc)
    dump("fuzz_vectors.json", {
        "text": "oracle_gen_text(seed, sigma, 0, n); periodic: np.resize(gen_text(seed,sigma,0,unit), n)",
        "pattern": "T[k:k+m], mutated: P[mut]=(P[mut]+1)%sigma",
        "rows": rows})


def deviations():
    rows = []
    # (1) EPSM tail miss: m=16, n=24, single occurrence at s=8 (epsm.c:330)
    T = po.gen_text(7, 128, 0, 24)
    P = T[8:24].copy()
    c = ref_counts(P, T)
    assert c["bf"] == 1 and c["epsm"] == 0, c
    rows.append({"name": "epsm_tail_miss", "seed": 7, "sigma": 128, "n": 24, "m": 16, "k": 8,
                 "truth": 1, "ref": {"epsm": 0}, "cite": "src/algos/epsm.c:330"})
    # larger instance of the same bug: m=32, n=4096, planted at s=n-m
    T = po.gen_text(8, 128, 0, 4096)
    P = T[4096 - 32:].copy()
    c = ref_counts(P, T)
    assert c["bf"] == 1 and c["epsm"] == 0, c
    rows.append({"name": "epsm_tail_miss_32", "seed": 8, "sigma": 128, "n": 4096, "m": 32, "k": 4064,
                 "truth": 1, "ref": {"epsm": 0}, "cite": "src/algos/epsm.c:330"})
    # (2) SO / BNDM search_large straddle: m=48, the first 40 bytes of P are the
    # last 40 bytes of T and the memory after T[n-1] holds P[40:48].  Truth 0;
    # the reference counts 1 because so.c:90 / bndm.c:101 read past T[n-1].
    T = po.gen_text(9, 128, 0, 200)
    P = np.concatenate([T[160:200], np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.uint8)])
    n, m = 200, 48
    buf = np.zeros(n + 64, dtype=np.uint8)
    buf[:n] = T
    buf[n:n + 8] = P[40:48]
    pb = np.zeros(m + 8, dtype=np.uint8)
    pb[:m] = P
    got = {}
    for a in ("so", "bndm", "bf", "hor"):
        got[a] = int(REF[a].lib.search(pb.ctypes.data, m, buf.ctypes.data, n))
    assert got["bf"] == 0 and got["hor"] == 0 and got["so"] == 1 and got["bndm"] == 1, got
    rows.append({"name": "so_bndm_straddle", "seed": 9, "sigma": 128, "n": 200, "m": 48,
                 "P_hex": P.tobytes().hex(), "tail_hex": P[40:48].tobytes().hex(),
                 "truth": 0, "ref": {"so": 1, "bndm": 1},
                 "cite": "src/algos/so.c:90, src/algos/bndm.c:101"})
    dump("deviations.json", {"rows": rows})


def english_vectors():
    src = "/root/reference/data/englishTexts/bible.txt"
    exc = os.path.join(OUT, "english_excerpt.txt")
    n = 262144
    with open(src, "rb") as f:
        data = f.read(n)
    with open(exc, "wb") as f:
        f.write(data)
    T = np.frombuffer(data, dtype=np.uint8)
    rows = []
    for m in (2, 4, 8, 16, 32, 64, 256, 1024):
        for k in (0, 12345, 131072, n - m - 1):
            P = T[k:k + m]
            c = ref_counts(P, T)
            assert len(set(c.values())) == 1, (m, k, c)
            rows.append({"m": m, "k": k, "count": c["bf"]})
    dump("english_vectors.json", {"text": "english_excerpt.txt = first 262144 bytes of data/englishTexts/bible.txt",
                                  "rows": rows})


if __name__ == "__main__":
    po.build(ref=True)
    survey_vectors()
    testc_cases()
    fuzz_vectors()
    deviations()
    english_vectors()
