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
  config5_vectors.json  BASELINE config 5's alphabets (sigma 2, 32, 256) x every length of src/sets.h:25
                        (2 .. 4096) on 2 MiB counter-based texts
  english_bible_world192.txt.xz + english_corpus_vectors.json
                        BASELINE config 4's unit: data/englishTexts as getText loads it through index.txt
                        (smart.c:95-138: bible.txt then world192.txt, 6,520,792 bytes), xz-compressed DATA,
                        and the reference's counts on it for every length of sets.h:25

    python tests/golden/gen_golden.py                 # everything
    python tests/golden/gen_golden.py config5 english_corpus   # only these
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


SETS_H_25 = (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)  # src/sets.h:25
MIN_M = {"raita": 2, "hash3": 3, "hash5": 5, "hash8": 8, "sbndm": 2}  # below: the reference returns -1 (raita.c:37, hash3.c:31, ...)


def agreed(c, m):
    """The one count all applicable reference algorithms returned (-1 exactly where the algorithm does not apply)."""
    vals = set()
    for a, v in c.items():
        if m < MIN_M.get(a, 1):
            assert v == -1, (a, m, v)
        elif v is not None:
            vals.add(v)
    assert len(vals) == 1, (m, c)
    return vals.pop()



def config5_vectors():
    """sigma in {2, 32, 256}, every length of sets.h:25, patterns cut at three offsets of a 2 MiB text
    (the last one ends 1 byte before the text does, as setOfRandomPatterns draws k < n-m, smart.c:153)."""
    n = (2 << 20) + 12345
    rows = []
    for sigma in (2, 32, 256):
        seed = 0xC0F5 + sigma
        T = po.gen_text(seed, sigma, 0, n)
        for m in SETS_H_25:
            for k in (4321, n // 2 + 7, n - m - 1):
                P = T[k:k + m].copy()
                rows.append({"sigma": sigma, "seed": seed, "n": n, "m": m, "k": k, "count": agreed(ref_counts(P, T), m)})
    dump("config5_vectors.json", {"text": "oracle_gen_text(seed, sigma, 0, n)", "pattern": "T[k:k+m]", "rows": rows})


def get_text(path, tsize):
    """getText of src/smart.c:95-138: the files named between '#' marks in <path>/index.txt, in order,
    concatenated and cut at tsize bytes."""
    out = bytearray()
    with open(os.path.join(path, "index.txt"), "rb") as f:
        idx = f.read()
    i = 0
    while i < len(idx) and len(out) < tsize:
        if idx[i:i + 1] == b"#":
            j = idx.index(b"#", i + 1)
            with open(os.path.join(path, idx[i + 1:j].decode()), "rb") as g:
                out += g.read(tsize - len(out))
            i = j + 1
        else:
            i += 1
    return bytes(out)


def english_corpus():
    import hashlib
    import lzma
    data = get_text("/root/reference/data/englishTexts", 1 << 30)
    assert len(data) == 6520792 and hashlib.md5(data[:4047392]).hexdigest() == "93fb92788b569c0387a50f4c99720ee7" \
        and hashlib.md5(data[4047392:]).hexdigest() == "30500a27cb7a15e6f2fa0032b06e06c3"
    with open(os.path.join(OUT, "english_bible_world192.txt.xz"), "wb") as f:
        f.write(lzma.compress(data, preset=9 | lzma.PRESET_EXTREME))
    T = np.frombuffer(data, dtype=np.uint8)
    n = len(T)
    rows = []
    for m in SETS_H_25:
        for k in (0, 12345, 524288, 4047392 - m // 2, n - m - 1):  # the fourth straddles the two files
            P = T[k:k + m].copy()
            rows.append({"m": m, "k": k, "count": agreed(ref_counts(P, T), m)})
    # SURVEY.md §8c starter table, english rows: the first 1,048,576 bytes (default -tsize, smart.c:416)
    n1 = 1048576
    want = {(2, 0): 52, (2, 12345): 6035, (2, 524288): 33904, (2, n1 - 3): 6248, (4, 0): 39, (4, 12345): 3952,
            (4, 524288): 1097, (4, n1 - 5): 1731, (8, 0): 1, (8, 12345): 125, (8, 524288): 18, (8, n1 - 9): 4,
            (32, 0): 1, (32, 12345): 3, (32, 524288): 1, (32, n1 - 33): 1, (256, 0): 1, (256, 12345): 1}
    prefix_rows = []
    for (m, k), cnt in sorted(want.items()):
        got = agreed(ref_counts(T[k:k + m].copy(), T[:n1]), m)
        assert got == cnt, (m, k, got, cnt)
        prefix_rows.append({"m": m, "k": k, "n": n1, "count": got})
    dump("english_corpus_vectors.json", {
        "text": "english_bible_world192.txt.xz = data/englishTexts/bible.txt + world192.txt (getText order), 6520792 bytes",
        "md5": hashlib.md5(data).hexdigest(), "n": n, "rows": rows,
        "survey_rows": prefix_rows, "survey_rows_text": "the first 1,048,576 bytes of the same text (SURVEY.md §8c table)"})


if __name__ == "__main__":
    po.build(ref=True)
    todo = sys.argv[1:] or ["survey", "testc", "fuzz", "deviations", "english", "config5", "english_corpus"]
    for name, fn in (("survey", survey_vectors), ("testc", testc_cases), ("fuzz", fuzz_vectors),
                     ("deviations", deviations), ("english", english_vectors), ("config5", config5_vectors),
                     ("english_corpus", english_corpus)):
        if name in todo:
            fn()
