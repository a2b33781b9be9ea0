"""GPU parity on BASELINE.json's configurations 3, 4 and 5 (through the C ABI, bit-exact):

* the reference's own counts (tests/golden/config5_vectors.json, english_corpus_vectors.json — written by
  gen_golden.py from oracle/_ref builds of src/algos/*.c) for every algorithm at every pattern length of
  src/sets.h:25 (2 .. 4096), on sigma 2 / 32 / 256 and on the English corpus, with the plan's own kernel
  choice and with every algorithm forced onto its own kernel (smartgpu_tune(0,1));
* at the configurations' full sizes (1 GiB, 4 GiB shards) through size-independent properties: all kernels
  agree, the sum over shards with an (m-1)-byte overlap equals the whole, a 32 MiB slice equals the oracle.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

import smart_amd  # noqa: E402
from smart_amd import Plan, Text, engine  # noqa: E402

ALGOS = smart_amd.ALGOS
CONFIG5 = ("hor", "bm", "kmp", "so", "epsm")  # the algorithms BASELINE config 5 names
SEED2 = 0x5EED0001
SETS_H_25 = (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    assert smart_amd.device_count() > 0, "no HIP device: " + smart_amd.lib().smartgpu_last_error().decode()


def applies(algo, m):
    return m >= smart_amd.MIN_M.get(algo, 1)


def counts(P, text, algos, **kw):
    return {a: smart_amd.search(a, P, text, **kw)[0] for a in algos if applies(a, len(P))}


def both_routings(fn):
    """fn() under the plan's own kernel choice and with every algorithm on its own kernel."""
    out = [fn()]
    engine.tune(0, 1)
    try:
        out.append(fn())
    finally:
        engine.tune(0, 0)
    return out


def test_config5_alphabets_every_length_against_the_reference(oracle):
    """sigma 2, 32, 256 x m = 2 .. 4096: the reference builds' counts, all sixteen algorithms, both routings."""
    g = load_golden("config5_vectors.json")
    texts = {}
    for r in g["rows"]:
        key = (r["sigma"], r["seed"], r["n"])
        if key not in texts:
            texts[key] = Text.generate(r["seed"], r["sigma"], r["n"])
            assert np.array_equal(texts[key].read(0, 4096), oracle.gen_text(r["seed"], r["sigma"], 0, 4096))
        text = texts[key]
        P = text.pattern(r["k"], r["m"])
        for got in both_routings(lambda: counts(P, text, ALGOS)):
            assert got and all(v == r["count"] for v in got.values()), (r, got)
    assert {r["sigma"] for r in g["rows"]} == {2, 32, 256} and {r["m"] for r in g["rows"]} == set(SETS_H_25)


def test_english_corpus_every_length_against_the_reference():
    """BASELINE config 4's unit (bible.txt || world192.txt as getText loads it): the reference builds' counts."""
    from smart_amd import corpus
    T = corpus.english_unit()
    g = load_golden("english_corpus_vectors.json")
    assert len(T) == g["n"] == 6520792
    text = Text.upload(T)
    for r in g["rows"]:
        P = T[r["k"]:r["k"] + r["m"]]
        for got in both_routings(lambda: counts(P, text, ALGOS)):
            assert got and all(v == r["count"] for v in got.values()), (r, got)
    for r in g["survey_rows"]:  # the first 1 MiB
        P = T[r["k"]:r["k"] + r["m"]]
        got = counts(P, text, ALGOS, off=0, n=r["n"])
        assert all(v == r["count"] for v in got.values()), (r, got)
    text.free()


def shard_sum(algo, P, text, n, parts=4):
    m = len(P)
    starts = n - m + 1
    total = 0
    for g in range(parts):
        a, b = starts * g // parts, starts * (g + 1) // parts
        total += smart_amd.search(algo, P, text, off=a, n=(b - a) + m - 1)[0]
    return total


def full_size_properties(oracle, text, n, host_slice, algos, ms, own_kernels_too=False):
    """All `algos` agree on the whole text, 4 shards sum to the whole, a 32 MiB slice equals the oracle."""
    sl = 32 << 20
    for j, m in enumerate(ms):
        k = oracle.splitmix64(0x0A77E2 + 4096 * j + m) % (n - m)
        P = text.pattern(k, m)
        runs = both_routings(lambda: counts(P, text, algos)) if own_kernels_too else [counts(P, text, algos)]
        for got in runs:
            assert len(set(got.values())) == 1 and got[algos[0]] >= 1, (m, got)
        whole = runs[0][algos[0]]
        for a in algos[:2]:
            assert shard_sum(a, P, text, n) == whole, (a, m)
        lo = max(0, min(k - (sl // 2), n - sl)) & ~4095
        want = oracle.search("hor", P, host_slice(lo, sl))
        for a in algos:
            if applies(a, m):
                assert smart_amd.search(a, P, text, off=lo, n=sl)[0] == want, (a, m, lo)


@pytest.mark.parametrize("sigma", [2, 32, 256])
def test_config5_shard_at_full_size(oracle, sigma):
    """One GPU's 4 GiB shard of BASELINE config 5 at each of its alphabets."""
    n = 1 << 32
    text = Text.generate(SEED2, sigma, n)
    full_size_properties(oracle, text, n, lambda lo, ln: text.read(lo, ln), CONFIG5, (2, 8, 32, 512, 4096))
    # a start position beyond 2^32 - m is not a start; the last window is
    P = text.pattern(n - 64, 64)
    assert len({smart_amd.search(a, P, text, off=n - 1000, n=1000)[0] for a in CONFIG5}) == 1
    text.free()


@pytest.mark.parametrize("sigma", [2, 32, 256])
def test_config5_at_32gib_on_one_gpu(oracle, sigma):
    """BASELINE config 5 at its STATED size — 2^35 bytes of rand-sigma — resident on ONE MI355X (288 GB).  The
    reference cannot hold such a text at all (`int TSIZE`, src/smart.c:416,454; `int n`, main.h:39), so parity at
    this size rests on the size-independent properties: the five named algorithms (five kernel families) agree on
    the whole text, the eight 4 GiB shards of the configuration with their (m-1)-byte overlap sum to the whole,
    32 MiB slices — around the planted pattern and straddling every multiple of 2^32 — equal the oracle, and the
    window cut at n-m is found.  Start offsets beyond 2^33 and (sigma 2, m 2) a count beyond 2^32 are exercised."""
    n = 1 << 35
    sl = 32 << 20
    text = Text.generate(SEED2, sigma, n)
    assert len(text) == n
    # the generator is counter-based: the device's bytes beyond 2^32 / 2^34 are the oracle's
    for lo in (0, (1 << 32) - 2048, (1 << 34) + 12345, n - 4096):
        assert np.array_equal(text.read(lo, 4096), oracle.gen_text(SEED2, sigma, lo, 4096)), lo
    for j, m in enumerate((2, 32, 4096)):
        k = (1 << 33) + oracle.splitmix64(0x0A77E2 + 4096 * j + m) % (n - (1 << 33) - m)  # planted beyond 2^33
        assert k >= 1 << 33
        P = text.pattern(k, m)
        got = counts(P, text, CONFIG5)
        assert len(set(got.values())) == 1 and got["hor"] >= 1, (sigma, m, got)
        whole = got["hor"]
        if sigma == 2 and m == 2:
            assert whole > 1 << 32  # a quarter of all start positions: the count itself needs 64 bits
        for a in ("hor", "kmp", "so"):  # tile kernels and the runs kernels cut the text differently
            assert shard_sum(a, P, text, n, parts=8) == whole, (a, sigma, m)
        # the planted copy and every 2^32 boundary inside a 32 MiB slice, against the oracle (two algorithms of the
        # oracle per slice: the restated Horspool and brute force, the definition of truth, bf.c:25-39)
        los = [(k - sl // 2) & ~4095] + [(b << 32) - sl // 2 - 4096 * b for b in range(1, 8)]
        for lo in los:
            lo = max(0, min(lo, n - sl))
            T = text.read(lo, sl)
            want = oracle.search("hor", P, T)
            assert want == oracle.search("bf", P, T)
            for a in CONFIG5:
                assert smart_amd.search(a, P, text, off=lo, n=sl)[0] == want, (a, sigma, m, lo)
        # a sub-range that starts beyond 2^34 and is not aligned to anything
        lo = (1 << 34) + 777
        T = text.read(lo, 1 << 20)
        assert {smart_amd.search(a, P, text, off=lo, n=1 << 20)[0] for a in CONFIG5} == {oracle.search("bf", P, T)}
    # the window cut at n - m is a start position, n - m + 1 is not
    for m in (2, 32, 4096):
        P = text.pattern(n - m, m)
        T = text.read(n - (1 << 20), 1 << 20)
        want = oracle.search("bf", P, T)
        assert want >= 1
        for a in CONFIG5:
            assert smart_amd.search(a, P, text, off=n - (1 << 20), n=1 << 20)[0] == want, (a, m)
            assert smart_amd.search(a, P, text, off=n - m, n=m)[0] == 1, (a, m)
            if m > 2:
                assert smart_amd.search(a, P, text, off=n - m + 1, n=m - 1)[0] == 0, (a, m)
    text.free()


@pytest.mark.parametrize("sigma", [4, 2])
def test_config3_at_full_size(oracle, sigma):
    """BASELINE config 3: Shift-Or and BNDM, m <= 64, 1 GiB of sigma 4 ("genome") and sigma 2 — on the plans'
    kernels and on bndm_scan / so_runs themselves."""
    n = 1 << 30
    text = Text.generate(SEED2, sigma, n)
    full_size_properties(oracle, text, n, lambda lo, ln: text.read(lo, ln), ("so", "bndm", "sa", "sbndm", "bndml", "kmp", "epsm"),
                         (2, 4, 8, 16, 32, 64), own_kernels_too=True)
    engine.tune(0, 1)
    try:
        pl = Plan("bndm", text.pattern(12345, 32))
        assert pl.kernel_name == "bndm_scan"
        pl.free()
    finally:
        engine.tune(0, 0)
    text.free()


def test_config4_english_tiled_to_4gib(oracle):
    """BASELINE config 4: the English unit tiled to exactly 2^32 bytes on the device, patterns from the first
    copy, m = 2 .. 4096.  Besides the properties above: a pattern that lies inside the unit occurs in every
    whole copy, so its count is at least floor(2^32 / unit) times its count in the unit."""
    from smart_amd import corpus
    unit = corpus.english_unit()
    n = 1 << 32
    text = Text.upload_tiled(unit, n)
    assert np.array_equal(text.read(len(unit) - 100, 200), np.concatenate([unit[-100:], unit[:100]]))

    def host_slice(lo, ln):
        idx = (np.arange(lo, lo + ln, dtype=np.int64)) % len(unit)
        return unit[idx]

    g = load_golden("english_corpus_vectors.json")
    in_unit = {(r["m"], r["k"]): r["count"] for r in g["rows"]}
    copies = n // len(unit)
    for j, m in enumerate(SETS_H_25):
        k = 12345
        P = unit[k:k + m]
        got = counts(P, text, ("bm", "hor", "kmp", "so", "epsm", "bndm"))
        assert len(set(got.values())) == 1, (m, got)
        assert got["bm"] >= copies * in_unit[(m, k)], (m, got, copies, in_unit[(m, k)])
        assert shard_sum("bm", P, text, n, parts=8) == got["bm"], m
        lo = (j * 509 * (1 << 20) + 4047392) % (n - (32 << 20)) & ~4095  # slices across copy boundaries
        want = oracle.search("bm", P, host_slice(lo, 32 << 20))
        for a in ("bm", "epsm", "kmp"):
            assert smart_amd.search(a, P, text, off=lo, n=32 << 20)[0] == want, (a, m)
    engine.tune(0, 1)  # bm_scan itself on English
    try:
        P = unit[12345:12345 + 64]
        pl = Plan("bm", P)
        assert pl.kernel_name == "bm_scan"
        pl.free()
        assert smart_amd.search("bm", P, text)[0] == smart_amd.search("epsm", P, text)[0]
    finally:
        engine.tune(0, 0)
    text.free()


@pytest.mark.parametrize("config,sigma,algo,m", [(2, 128, "hor", 32), (3, 4, "so", 16), (5, 2, "kmp", 8), (3, 2, "bndm", 32),
                                                 (5, 256, "bm", 512), (4, "english", "bm", 128)])
def test_whole_text_against_the_cpu_at_1gib(oracle, config, sigma, algo, m):
    """One pattern per configuration over the WHOLE 1 GiB text on both sides: the GPU count (the plan's kernel and
    the algorithm's own) against the CPU restatement of the same algorithm run over all of the text (split by core
    with an (m-1)-byte overlap).  Full-size parity does not rest on slices and kernel agreement alone."""
    import os
    n = 1 << 30
    if sigma == "english":
        from smart_amd import corpus
        unit = corpus.english_unit()
        text = Text.upload_tiled(unit, n)
        P = unit[12345:12345 + m].copy()
    else:
        text = Text.generate(SEED2, sigma, n)
        P = text.pattern(oracle.splitmix64(0x0A77E2 + m) % (n - m), m)
    T = text.read(0, n)
    want = oracle.search(algo, P, T, threads=os.cpu_count() or 1)
    assert want >= 1
    got = both_routings(lambda: smart_amd.search(algo, P, text)[0])
    assert got == [want, want], (config, sigma, algo, m, got, want)
    # the other end of the contract: truth is brute force (bf.c:25-39)
    assert oracle.search("bf", P, T, threads=os.cpu_count() or 1) == want
    del T
    text.free()
