"""GPU parity tests: every HIP kernel, called through the C ABI, against the
oracle and the golden vectors.  Bit-exact (integer counts)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, fuzz_case, load_golden

pytestmark = pytest.mark.gpu

import smart_amd  # noqa: E402
from smart_amd import Plan, Text  # noqa: E402

ALGOS = smart_amd.ALGOS
SEED2 = 0x5EED0001  # BASELINE config 2 corpus seed


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    assert smart_amd.device_count() > 0, "no HIP device: " + smart_amd.lib().smartgpu_last_error().decode()


def applies(algo, m):
    """raita.c:37 returns -1 ("not applicable") for m < 2, hash3/5/8.c for m < 3/5/8; every other
    algorithm takes any m >= 1."""
    return m >= smart_amd.MIN_M.get(algo, 1)


def gpu_counts(P, text, algos=ALGOS, **kw):
    """Counts of every algorithm that applies to this pattern length; an algorithm that does not
    must say so (SMARTGPU_NA) instead of counting."""
    out = {}
    for a in algos:
        if applies(a, len(P)):
            out[a] = smart_amd.search(a, P, text, **kw)[0]
        else:
            with pytest.raises(smart_amd.SmartGpuError):
                smart_amd.search(a, P, text, **kw)
    return out


def test_testc_cases():
    """src/test.c:252-382 through the reference's own `int search(P,m,T,n)` shape."""
    for r in load_golden("testc_cases.json")["rows"]:
        P, T = r["P"].encode(), r["T"].encode()
        for a in ALGOS:
            assert smart_amd.search_host(a, P, T) == (r["count"] if applies(a, len(P)) else -1), (a, r)


def test_fuzz_vectors(oracle):
    rows = load_golden("fuzz_vectors.json")["rows"]
    for r in rows:
        P, T = fuzz_case(oracle, r)
        text = Text.upload(T)
        got = gpu_counts(P, text)
        text.free()
        for a in got:
            assert got[a] == r["count"], (a, r, got)


def test_fuzz_vectors_every_algorithm_on_its_own_kernel(oracle):
    """The plans send short patterns and patterns whose symbols repeat to so_runs (api.cpp build_blob); with
    smartgpu_tune(0,1) every algorithm counts on its own kernel — hor_scan, bm_scan, bndm_scan, bndml_scan,
    hor_scan_bp — for every fuzz vector (small alphabets, periodic texts, m = 1 .. 300), and with (0,3) the skip
    algorithms count on the packed matcher."""
    from smart_amd import engine
    rows = load_golden("fuzz_vectors.json")["rows"]
    for setting, algos, step in ((1, ALGOS, 1), (3, ("hor", "bm", "bndm", "qs", "raita", "hash3", "sbndm", "bndml", "tunedbm"), 3)):
        engine.tune(0, setting)
        try:
            for r in rows[::step]:
                P, T = fuzz_case(oracle, r)
                text = Text.upload(T)
                got = gpu_counts(P, text, algos=algos)
                text.free()
                for a in got:
                    assert got[a] == r["count"], (setting, a, r, got)
            if setting == 1:
                P = oracle.gen_text(3, 4, 0, 64)
                assert smart_amd.kernel_for("bm", P) == "bm_scan" and smart_amd.kernel_for("hor", P[:5]) == "hor_scan"
        finally:
            engine.tune(0, 0)


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_horspool_variants(oracle, ab_library, variant):
    """All Horspool regimes (flat LDS tile / bank-private layout / packed) on the fuzz
    vectors, the dense small-alphabet case and sub-ranges."""
    from smart_amd import engine
    engine.tune(0, variant)
    try:
        for r in load_golden("fuzz_vectors.json")["rows"]:
            P, T = fuzz_case(oracle, r)
            text = Text.upload(T)
            got = smart_amd.search("hor", P, text)[0]
            text.free()
            assert got == r["count"], (variant, r, got)
        T = oracle.gen_text(77, 2, 0, 1_500_000)
        text = Text.upload(T)
        for m in (1, 2, 3, 4, 7, 8, 15, 16, 33, 100, 255):
            P = T[4321:4321 + m]
            assert smart_amd.search("hor", P, text)[0] == oracle.search("epsm", P, T), (variant, m)
            off, n = 70001, 999999
            assert smart_amd.search("hor", P, text, off=off, n=n)[0] == oracle.search("epsm", P, T[off:off + n]), (variant, m)
    finally:
        engine.tune(0, 0)


def test_skip_loops_forced_for_short_patterns(oracle):
    """m <= 16 normally goes to the packed matcher; force the LDS-tile skip loops of
    BM and BNDM (and Karp-Rabin's own kernel) so that they stay covered for short patterns too."""
    from smart_amd import engine
    engine.tune(0, 1)
    try:
        for r in load_golden("fuzz_vectors.json")["rows"]:
            if r["m"] > 17:
                continue
            P, T = fuzz_case(oracle, r)
            text = Text.upload(T)
            got = gpu_counts(P, text, algos=("bm", "bndm", "kr"))  # kr: its own kernel (m < 8: the subtracting rolling hash)
            text.free()
            assert got["bm"] == r["count"] and got["bndm"] == r["count"] and got["kr"] == r["count"], (r, got)
    finally:
        engine.tune(0, 0)


@pytest.mark.parametrize("policy", [1, 3])
def test_packed_load_policies(oracle, ab_library, policy):
    """The packed matcher's alternative data paths (both loads cached / one
    non-temporal load + cross-lane shuffle) give the same counts."""
    from smart_amd import engine
    engine.tune(7, policy)
    try:
        for r in load_golden("fuzz_vectors.json")["rows"][::2]:
            P, T = fuzz_case(oracle, r)
            text = Text.upload(T)
            got = gpu_counts(P, text, algos=("epsm", "hor"))
            text.free()
            assert got["epsm"] == r["count"] and got["hor"] == r["count"], (policy, r, got)
        T = oracle.gen_text(5, 2, 0, 3_000_000)
        text = Text.upload(T)
        for m in (1, 3, 8, 16, 17, 100):
            P = T[999:999 + m]
            want = oracle.search("epsm", P, T)
            assert smart_amd.search("epsm", P, text)[0] == want, (policy, m)
            assert smart_amd.search("epsm", P, text, off=1234567, n=1000001)[0] == oracle.search("epsm", P, T[1234567:2234568]), (policy, m)
    finally:
        engine.tune(7, 0)


def test_packed_matcher_sums_of_absolute_differences(oracle):
    """packed_scan's v_mqsad_pk_u16_u8 modes (k_packed.hip, modes 5-9: epsm.c:165-223's mpsadbw filter on this machine) against
    its dword compares (tune(7,9)) and the oracle: every length 1..20 and some beyond, on two, four, 128 and 256 byte values;
    patterns that hold the byte values 0, 1, 2, ... (the XOR byte K must dodge them: a reference byte of 0 is skipped by the
    instruction); dense and planted occurrences; sub-ranges whose first and last rows are partial; tune(7,6) = these modes
    on every text and at every length."""
    from smart_amd import engine
    rng = np.random.default_rng(77)
    n = 2_000_003
    for sigma in (2, 4, 128, 256):
        T = oracle.gen_text(4242 + sigma, sigma, 0, n)
        if sigma == 256:
            T[100_000:100_040] = np.arange(40, dtype=np.uint8)  # a stretch 0, 1, 2, ...: patterns cut from it hold every small value
        text = Text.upload(T)
        for m in list(range(1, 21)) + [33, 100, 1000]:
            cuts = [T[777:777 + m].copy(), T[n - m:].copy()]
            if sigma == 256:
                cuts.append(T[100_000:100_000 + min(m, 40)].copy())
            for P in cuts:
                for k in rng.integers(0, n - len(P), 50):
                    T[k:k + len(P)] = P
            text.free()
            text = Text.upload(T)
            for P in cuts:
                want = oracle.search("bf", P, T)
                sub = oracle.search("bf", P, T[1_234_567 - 1:1_234_567 - 1 + 500_001])
                got = {}
                for t in (0, 6, 9):
                    engine.tune(7, t)
                    try:
                        got[t] = (smart_amd.search("epsm", P, text)[0], smart_amd.search("epsm", P, text, off=1_234_566, n=500_001)[0])
                    finally:
                        engine.tune(7, 0)
                assert all(v == (want, sub) for v in got.values()), (sigma, len(P), got, want, sub)
                if len(P) <= 7:  # the short patterns of the skip algorithms take the same road
                    assert smart_amd.search("hor", P, text)[0] == want and smart_amd.search("bndm", P, text)[0] == want, (sigma, len(P))
        text.free()


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_alternate_serial_kernels(oracle, ab_library, variant):
    """SO and KMP normally run on the bank-private / full-table runs kernels; the variants kept
    for A/B measurements must give the same counts: 1 = LDS tiles (so_scan, kmp_scan),
    2 = shared-table so_runs and the failure-link kmp_links_runs, 3 = Shift-And in its own AND
    form (by default it counts in the complemented, Shift-Or form) and kmp_runs1 (running maximum, half-line
    loader), 4 = so_runs1 (a step per byte)."""
    from smart_amd import engine
    engine.tune(6, variant)
    engine.tune(3, min(variant, 3))
    try:
        for r in load_golden("fuzz_vectors.json")["rows"][::2]:
            P, T = fuzz_case(oracle, r)
            text = Text.upload(T)
            got = gpu_counts(P, text, algos=("so", "kmp", "sa"))
            text.free()
            assert got["so"] == r["count"] and got["kmp"] == r["count"] and got["sa"] == r["count"], (r, got)
    finally:
        engine.tune(6, 0)
        engine.tune(3, 0)


def test_survey_vectors(oracle):
    g = load_golden("survey_vectors.json")
    texts = {}
    for r in g["rows"]:
        if r["sigma"] not in texts:
            texts[r["sigma"]] = (Text.upload(oracle.textgen(r["sigma"], r["n"])),)
        text = texts[r["sigma"]][0]
        P = text.pattern(r["k"], r["m"])
        got = gpu_counts(P, text)
        for a in got:
            assert got[a] == r["count"], (a, r, got)


def test_english_vectors():
    T = np.fromfile(os.path.join(GOLDEN, "english_excerpt.txt"), dtype=np.uint8)
    text = Text.upload(T)
    for r in load_golden("english_vectors.json")["rows"]:
        got = gpu_counts(T[r["k"]:r["k"] + r["m"]], text)
        for a in got:
            assert got[a] == r["count"], (a, r, got)


def test_documented_deviations(oracle):
    """Truth (bf.c semantics) where the reference itself deviates: EPSM tail
    miss (epsm.c:330) and the SO/BNDM straddle over-read (so.c:90, bndm.c:101)."""
    for r in load_golden("deviations.json")["rows"]:
        T = oracle.gen_text(r["seed"], r["sigma"], 0, r["n"])
        P = np.frombuffer(bytes.fromhex(r["P_hex"]), dtype=np.uint8) if "P_hex" in r else T[r["k"]:r["k"] + r["m"]]
        if "tail_hex" in r:
            # put the bytes that fool the reference right after the searched range
            tail = np.frombuffer(bytes.fromhex(r["tail_hex"]), dtype=np.uint8)
            text = Text.upload(np.concatenate([T, tail]))
            got = gpu_counts(P, text, off=0, n=r["n"])
        else:
            text = Text.upload(T)
            got = gpu_counts(P, text)
        for a in got:
            assert got[a] == r["truth"], (a, r, got)


def test_edges_and_ranges(oracle):
    T = oracle.gen_text(11, 4, 0, 70000)
    text = Text.upload(T)
    # m == n, m > n, n == 0, single byte
    for a in ALGOS:
        assert smart_amd.search(a, T[:100], text, off=0, n=100)[0] == 1
        assert smart_amd.search(a, T[:100], text, off=5, n=50)[0] == 0
        if applies(a, 3):
            assert smart_amd.search(a, T[:3], text, off=17, n=0)[0] == 0
        if applies(a, 1):
            assert smart_amd.search(a, T[9:10], text, off=9, n=1)[0] == 1
    # arbitrary sub-ranges (shard-style): count == oracle on the slice
    rng = np.random.default_rng(5)
    for _ in range(40):
        off = int(rng.integers(0, 60000))
        n = int(rng.integers(1, 70000 - off))
        m = int(rng.choice([1, 2, 5, 16, 33, 70, 300]))
        k = int(rng.integers(0, 70000 - m))
        P = T[k:k + m]
        want = oracle.search("bf", P, T[off:off + n])
        got = gpu_counts(P, text, off=off, n=n)
        for a in got:
            assert got[a] == want, (a, off, n, m, k, got, want)
    # errors: m = 0, m > XSIZE, bad range
    for bad in (T[:0], np.zeros(4201, dtype=np.uint8)):
        with pytest.raises(smart_amd.SmartGpuError):
            smart_amd.search("hor", bad, text)
    with pytest.raises(smart_amd.SmartGpuError):
        smart_amd.search("hor", T[:4], text, off=69999, n=5)


def test_large_patterns_and_periodic_text(oracle):
    """m up to XSIZE (sets.h:25 goes to 4096), halo > LDS halo, dense overlaps."""
    T = oracle.gen_text(21, 128, 0, 300000)
    text = Text.upload(T)
    for m in (257, 258, 512, 1024, 4096, 4200):
        for k in (0, 12345, 300000 - m):
            P = T[k:k + m]
            got = gpu_counts(P, text)
            for a in got:
                assert got[a] == 1, (a, m, k, got)
    text.free()
    A = np.full(100000, ord("a"), dtype=np.uint8)
    text = Text.upload(A)
    for m in (1, 2, 31, 32, 33, 64, 300, 1000):
        want = 100000 - m + 1
        got = gpu_counts(A[:m], text)
        for a in got:
            assert got[a] == want, (a, m, got)
    AB = np.resize(np.frombuffer(b"ab", dtype=np.uint8), 65536 + 7)
    text = Text.upload(AB)
    for m in (2, 3, 40, 41, 600):
        want = oracle.search("bf", AB[:m], AB)
        got = gpu_counts(AB[:m], text)
        for a in got:
            assert got[a] == want, (a, m, got)


def test_occurrences_across_tile_and_lane_boundaries(oracle):
    """Occurrences planted so that they start, end and straddle the kernels' work boundaries:
    16 KiB tiles, 64-byte lane segments, 2 KiB runs, 4 KiB packed rows (the forward byte of
    Quick Search, the back halo of the hash and rolling-hash variants, the run restarts)."""
    rng = np.random.default_rng(2024)
    n = 5 * 16384 + 777
    for m in (2, 3, 8, 17, 33, 64, 100, 300):
        T = rng.integers(0, 250, n, dtype=np.uint8)
        P = rng.integers(0, 250, m, dtype=np.uint8)
        P[-1] = 251  # bytes outside the text's alphabet: only planted copies match
        P[0] = 252
        spots = []
        for edge in (2048, 4096, 16384, 32768, 49152, 65536):
            for d in (-m - 1, -m, -m + 1, -1, 0, 1, 63 - m, 64 - m):
                k = edge + d
                if 0 <= k and k + m <= n and all(abs(k - q) >= m for q in spots):
                    spots.append(k)
        spots += [0, n - m]
        for k in spots:
            T[k:k + m] = P
        want = oracle.search("bf", P, T)
        assert want == len(set(spots))
        text = Text.upload(T)
        got = gpu_counts(P, text)
        text.free()
        for a in got:
            assert got[a] == want, (a, m, got[a], want)


def test_small_alphabets_dense_matches(oracle):
    """BASELINE config 3 regime: sigma 2 and 4, m <= 64, many occurrences."""
    for sigma in (2, 4):
        T = oracle.gen_text(31 + sigma, sigma, 0, 2_000_000)
        text = Text.upload(T)
        for m in (1, 2, 4, 8, 16, 32, 64):
            P = T[777:777 + m]
            want = oracle.search("epsm", P, T)
            got = gpu_counts(P, text)
            for a in got:
                assert got[a] == want, (a, sigma, m, got, want)


def test_two_symbol_patterns_count_on_shift_or_runs_and_on_their_own_kernels(oracle):
    """A pattern of 16+ bytes over two or three symbols is counted by so_runs whatever the algorithm
    (api.cpp build_blob / launch.hip launch_scan); tune(0,1) keeps every algorithm on its own kernel.
    Both must give the oracle's count; patterns over larger alphabets are not rerouted."""
    from smart_amd import engine
    own = ("kmp", "kr")
    for sigma, n in ((2, 1_500_000), (3, 1_000_003)):
        T = oracle.gen_text(977 + sigma, sigma, 0, n)
        text = Text.upload(T)
        for m in (16, 17, 31, 32, 33, 64, 100, 300, 4096):
            for P in (T[4321:4321 + m].copy(), np.resize(T[9:11], m)):
                want = oracle.search("hor", P, T)
                for a in ALGOS:
                    if len(set(P.tolist())) > 2 or not applies(a, m):
                        continue
                    pl = Plan(a, P)
                    if a == "bndm" or (a == "bndml" and m <= 32):  # round 3: bndm_scan reads 8 bytes of a 32-byte window per step;
                        # round 4: two symbols, 8+ bytes: its gram form at any length
                        assert pl.kernel_name == ("bndm_scan" if m >= 32 or len(set(P.tolist())) == 2 else "so_runs"), (a, m, pl.kernel_name)
                    elif a in ("hor", "bm", "tunedbm"):  # round 4: Horspool, Tuned BM and Boyer-Moore on grams (two symbols: 32+ bytes)
                        assert pl.kernel_name == (("bm_scan" if a == "bm" else "hor_scan") if len(set(P.tolist())) == 2 and m >= 32 else "so_runs"), (a, m, pl.kernel_name)
                    else:
                        assert (pl.kernel_name == "so_runs") == (a not in own), (a, m, pl.kernel_name)
                    pl.free()
                got = gpu_counts(P, text)
                assert all(v == want for v in got.values()), (sigma, m, got, want)
                if m in (17, 100, 4096):  # a shard-style sub-range through the rerouted plans
                    off, nn = 123_457, 700_001
                    sub = oracle.search("hor", P, T[off:off + nn])
                    for a in ("hor", "bm", "bndm", "epsm", "qs", "bndml"):
                        assert smart_amd.search(a, P, text, off=off, n=nn)[0] == sub, (a, sigma, m)
                engine.tune(0, 1)
                try:
                    pl = Plan("bm", P)
                    assert pl.kernel_name == "bm_scan"
                    pl.free()
                    got = gpu_counts(P, text)
                finally:
                    engine.tune(0, 0)
                assert all(v == want for v in got.values()), ("own kernels", sigma, m, got, want)
        text.free()
    T = oracle.gen_text(5, 128, 0, 100_000)
    for a in ("hor", "bm", "bndm", "epsm", "qs"):
        pl = Plan(a, T[50:82])
        assert pl.kernel_name != "so_runs", a
        pl.free()


def test_device_generator_matches_oracle(oracle):
    for sigma, off, n in ((128, 0, 100000), (2, 12345, 70001), (250, 7, 4099), (256, 8, 64), (4, 3, 1)):
        text = Text.generate(SEED2, sigma, n, off=off)
        assert np.array_equal(text.read(0, n), oracle.gen_text(SEED2, sigma, off, n)), (sigma, off, n)
    unit = oracle.gen_text(1, 128, 0, 1000)
    text = Text.upload_tiled(unit, 5555, phase=37)
    assert np.array_equal(text.read(0, 5555), np.resize(np.roll(unit, -37), 5555))


def test_plan_slots_and_timing(oracle):
    T = oracle.gen_text(41, 128, 0, 1 << 20)
    text = Text.upload(T)
    pats = [T[k:k + 32] for k in (5, 70000, 999000)]
    plans = [Plan("hor", p) for p in pats]
    for i, p in enumerate(plans):
        p.launch(text, slot=i, timed=True)
    for i, p in enumerate(plans):
        c, ms = p.result(i)
        assert c == oracle.search("hor", pats[i], T)
        assert ms > 0
    assert plans[0].kernel_name == "hor_scan"  # m=32: flat-tile regime


def test_full_size_properties(oracle):
    """BASELINE config 2 at full size (1 GiB rand128, m in {4,8,32,256}) through
    size-independent properties: all six kernels agree; shard sums with an (m-1)
    overlap equal the whole; a 32 MiB slice equals the oracle."""
    n = 1 << 30
    text = Text.generate(SEED2, 128, n)
    for j, m in enumerate((4, 8, 32, 256)):
        k = oracle.splitmix64(0x0A77E2 + 4096 * j + m) % (n - m)
        P = text.pattern(k, m)
        got = gpu_counts(P, text)
        assert len(set(got.values())) == 1 and got["hor"] >= 1, (m, got)
        # 4 shards by start offset, each seeing m-1 extra bytes
        starts = n - m + 1
        total = 0
        for g in range(4):
            a, b = starts * g // 4, starts * (g + 1) // 4
            total += smart_amd.search("hor", P, text, off=a, n=(b - a) + m - 1)[0]
        assert total == got["hor"], (m, total, got)
        # slice around the planted occurrence against the oracle
        lo = max(0, min(k - (16 << 20), n - (32 << 20)))
        sl = text.read(lo, 32 << 20)
        want = oracle.search("hor", P, sl)
        for a in ("hor", "epsm", "so"):
            assert smart_amd.search(a, P, text, off=lo, n=32 << 20)[0] == want, (a, m)


def test_text_beyond_4gib():
    """64-bit offsets: a 5 GiB text (BASELINE configs 4-5 put 4 GiB on each GPU),
    patterns cut at the very end, on both sides of 2^32 and at the start; the six
    kernels must agree, find the planted occurrence, and a sub-range straddling
    2^32 must see exactly the occurrences inside it."""
    n = 5 * (1 << 30) + 12345
    text = Text.generate(SEED2, 128, n)
    lo, ln = (1 << 32) - 100000, 200000
    for m in (4, 40, 4096):
        for k in (n - m, (1 << 32) - 5, (1 << 32) + 11, 17):
            P = text.pattern(k, m)
            got = gpu_counts(P, text)
            assert len(set(got.values())) == 1 and got["hor"] >= 1, (m, k, got)
            sub = gpu_counts(P, text, algos=("hor", "kmp", "so", "epsm"), off=lo, n=ln)
            inside = 1 if (lo <= k and k + m <= lo + ln) else 0
            assert len(set(sub.values())) == 1 and sub["hor"] >= inside, (m, k, sub)
            if m >= 40:
                assert sub["hor"] == inside, (m, k, sub)


def test_multi_gpu_text_in_one_process(oracle):
    """smartgpu_mtext_*: shards by start offset with an overlap, per-shard searches, one
    reduction.  On a one-GPU box the shard arithmetic is exercised by listing device 0
    several times (host reduce); the RCCL all-reduce path runs with one rank."""
    from smart_amd import MultiText
    T = oracle.gen_text(99, 4, 0, 1_000_003)
    for k in (1, 2, 3, 8):
        mt = MultiText.upload(T, k, devices=[0] * k)
        assert len(mt) == len(T)
        for m in (1, 2, 31, 300, 4200):
            P = T[333331:333331 + m]
            want = oracle.search("epsm", P, T)
            for algo in ("hor", "kmp", "so"):
                assert mt.search(algo, P, reduce="host")[0] == want, (k, m, algo)
        mt.free()
    mt = MultiText.generate(SEED2, 128, 50_000_000, 1)
    ref = Text.generate(SEED2, 128, 50_000_000)
    P = ref.pattern(40_000_000, 64)
    c, pre_ms, run_ms = mt.search("bm", P, reduce="rccl")
    assert c == smart_amd.search("bm", P, ref)[0] >= 1 and run_ms > 0
    # a pattern that straddles a shard boundary is owned by exactly one shard
    n = 3_000_000
    mt = MultiText.generate(SEED2, 128, n, 3, devices=[0, 0, 0])
    ref = Text.generate(SEED2, 128, n)
    for m in (2, 100, 4000):
        P = ref.pattern(1_000_000 - m // 2, m)  # shard 0/1 boundary is at 1,000,000
        assert mt.search("hor", P, reduce="host")[0] == smart_amd.search("hor", P, ref)[0] >= 1


def positions_by_definition(P, T):
    """Every s with T[s..s+m) == P (bf.c semantics), by numpy."""
    m, n = len(P), len(T)
    if m > n:
        return np.empty(0, dtype=np.uint64)
    ok = np.ones(n - m + 1, dtype=bool)
    for i in range(m):
        ok &= T[i:n - m + 1 + i] == P[i]
        if not ok.any():
            break
    return np.flatnonzero(ok).astype(np.uint64)


def test_occurrence_positions(oracle):
    """smartgpu_find64 (extension): the positions, ascending, equal the
    definition's; their number equals every counting kernel's."""
    for sigma, n in ((2, 300000), (4, 300000), (128, 2_000_000)):
        T = oracle.gen_text(77 + sigma, sigma, 0, n)
        text = Text.upload(T)
        for m in (1, 2, 3, 4, 7, 8, 12, 16, 17, 33, 100):
            P = T[4321:4321 + m]
            want = positions_by_definition(P, T)
            pos, cnt = smart_amd.find(P, text, cap=max(len(want), 1))
            assert cnt == len(want) == smart_amd.search("epsm", P, text)[0], (sigma, m)
            assert np.array_equal(pos, want), (sigma, m)
            # a sub-range: positions stay relative to text byte 0
            off, ln = 1000, 150000
            sub = want[(want >= off) & (want + m <= off + ln)]
            pos, cnt = smart_amd.find(P, text, off=off, n=ln, cap=max(len(want), 1))
            assert cnt == len(sub) and np.array_equal(pos, sub), (sigma, m)
        text.free()
    # dense: every offset of a unary text; and a buffer that is too small
    A = np.full(100000, ord("a"), dtype=np.uint8)
    text = Text.upload(A)
    for m in (1, 5, 16, 40):
        pos, cnt = smart_amd.find(A[:m], text, cap=100000)
        assert cnt == 100000 - m + 1 and np.array_equal(pos, np.arange(cnt, dtype=np.uint64)), m
    pos, cnt = smart_amd.find(A[:5], text, cap=10)
    assert pos is None and cnt == 100000 - 5 + 1
    pos, cnt = smart_amd.find(b"zz", text, cap=4)
    assert cnt == 0 and len(pos) == 0
    text.free()


def test_pattern_set_in_one_call(oracle):
    """smartgpu_search_batch64 / smartgpu_msearch_batch64: the harness loop of smart.c:312-345 as one call —
    K patterns over one resident text, tables in one arena, launches back to back, one read-back of K counts.
    Counts must equal the per-call path and the oracle; every algorithm; table sets larger than the staging
    buffer (KMP: 135 KB per pattern) go up in several copies."""
    from smart_amd import MultiText
    n = 1 << 20  # SMART's stock text size (smart.c:416)
    T = oracle.gen_text(SEED2, 128, 0, n)
    text = Text.upload(T)
    rng = np.random.default_rng(11)
    for m, K in ((32, 40), (2, 9), (8, 17), (300, 5), (4096, 3)):
        pats = [T[k:k + m].copy() for k in rng.integers(0, n - m, K)]
        want = [oracle.search("bf", p, T) for p in pats]
        for a in ALGOS:
            if not applies(a, m):
                with pytest.raises(smart_amd.SmartGpuError):
                    smart_amd.search_batch(a, pats, text)
                continue
            counts, pre, run, batch_ms = smart_amd.search_batch(a, pats, text)
            assert counts.tolist() == want, (a, m, counts.tolist(), want)
            assert (pre > 0).all() and (run > 0).all() and batch_ms > 0, (a, m)
            counts2, _, run2, _ = smart_amd.search_batch(a, pats, text, per_pattern_times=False)
            assert counts2.tolist() == want and run2 is None
        # a sub-range, as a shard would search it
        off, nn = 123_457, 500_001
        sub = [oracle.search("bf", p, T[off:off + nn]) for p in pats]
        for a in ("hor", "kmp", "so", "epsm"):
            assert smart_amd.search_batch(a, pats, text, off=off, n=nn)[0].tolist() == sub, (a, m)
    # 560 KMP patterns of 200 bytes (full 64 KB tables): 39 MB of blobs through the 32 MB staging buffer in two copies;
    # the arena grows once
    big = [T[k:k + 200].copy() for k in rng.integers(0, n - 200, 560)]
    assert smart_amd.search_batch("kmp", big, text)[0].tolist() == [oracle.search("bf", p, T) for p in big]
    pats = [T[k:k + 32].copy() for k in rng.integers(0, n - 32, 300)]
    want = [oracle.search("bf", p, T) for p in pats]
    assert smart_amd.search_batch("kmp", pats, text)[0].tolist() == want
    assert smart_amd.search_batch("hor", pats, text)[0].tolist() == want
    with pytest.raises(smart_amd.SmartGpuError):  # one length per set (the harness loops per length)
        smart_amd.search_batch("hor", [T[:4], T[:5]], text)
    # several shards, one reduction of the K counts
    for k in (1, 3):
        mt = MultiText.upload(T, k, devices=[0] * k)
        counts, pre, batch_ms = mt.search_batch("bm", pats[:50], reduce="host")
        assert counts.tolist() == want[:50] and batch_ms > 0, k
        mt.free()
    mt = MultiText.upload(T, 1)
    assert mt.search_batch("so", pats[:20], reduce="rccl")[0].tolist() == want[:20]
    mt.free()
    text.free()


def test_small_texts_on_the_runs_kernels(oracle):
    """Texts too small to give every wave 2 KiB runs get shorter runs (kernels.hip balanced_run_len): counts on
    1 KiB .. 3 MiB texts, every m, sub-ranges that start and end inside runs."""
    for sigma, n in ((128, 1 << 20), (4, 300_001), (2, 70_000), (128, 1025), (16, 3 << 20)):
        T = oracle.gen_text(4242 + n, sigma, 0, n)
        text = Text.upload(T)
        for m in (1, 2, 3, 16, 29, 30, 33, 64, 254, 255, 256, 1000):
            if m >= n:
                continue
            k = (n // 3) & ~1
            P = T[k:k + m]
            want = oracle.search("bf", P, T)
            got = gpu_counts(P, text, algos=("kmp", "so", "sa"))
            assert all(v == want for v in got.values()), (sigma, n, m, got, want)
            off, nn = n // 7 + 3, n // 2 + 1
            sub = oracle.search("bf", P, T[off:off + nn])
            got = gpu_counts(P, text, algos=("kmp", "so", "sa"), off=off, n=nn)
            assert all(v == sub for v in got.values()), (sigma, n, m, got, sub)
        text.free()


def test_runs_kernels_where_the_text_does_not_fill_the_runs(oracle):
    """Lengths that leave the last run of kmp_runs / so_runs with one start position, half of them, or the last
    group of 64 runs partly empty, ranges that start inside a run, and run lengths that the balancing has to round
    (kernels.hip balanced_run_len, first_group; tools/nvar_probe.py times the same cases)."""
    n = (1 << 22) + 9000
    T = oracle.gen_text(SEED2 + 5, 4, 0, n)  # a small alphabet: the ends of the runs see occurrences
    text = Text.upload(T)
    rng = np.random.default_rng(3)
    for m in (5, 8, 62, 63, 200, 255, 300, 4096):
        P = T[777_000:777_000 + m].copy()
        for sub in ((1 << 22) + m - 1, (1 << 22) + m - 1 - 127, (1 << 22) - 1, (1 << 22) + 1, 4_000_001, 3_333_333, 1_000_000 + m, 65_537):
            for off in (0, 1, 4097, 123_457):
                nn = min(sub, n - off)
                want = oracle.search("bf", P, T[off:off + nn])
                for a in ("kmp", "so", "sa"):
                    assert smart_amd.search(a, P, text, off=off, n=nn)[0] == want, (a, m, sub, off)
    # a pattern whose own end is the text's end, and one byte less of text
    for m in (8, 300):
        P = T[n - m:].copy()
        for a in ("kmp", "so"):
            assert smart_amd.search(a, P, text)[0] == oracle.search("bf", P, T)
            assert smart_amd.search(a, P, text, off=0, n=n - 1)[0] == oracle.search("bf", P, T[:n - 1])
    text.free()


def test_pattern_set_larger_than_one_grid(oracle):
    """A set with more patterns than gridDim.y holds (65535) on a small text: the one-grid form launches the
    group in slices; every count equals the definition.  K beyond the documented 2^18 is an argument error."""
    n = 4096
    T = oracle.gen_text(5, 4, 0, n)
    text = Text.upload(T)
    K, m = 70_001, 4
    pats = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(np.resize(T, K + m), m)[:K])
    # the 256 possible patterns of 4 symbols over sigma 4, counted by definition
    win = np.lib.stride_tricks.sliding_window_view(T, m)
    code = lambda a: (a.astype(np.int64) * np.array([64, 16, 4, 1])).sum(axis=-1)  # noqa: E731
    hist = np.bincount(code(win), minlength=256)
    want = hist[code(pats)]
    for algo in ("hor", "so", "epsm"):
        got = smart_amd.search_batch(algo, list(pats), text, per_pattern_times=False)[0]
        assert np.array_equal(got.astype(np.int64), want), algo
    with pytest.raises(smart_amd.engine.SmartGpuError):
        smart_amd.search_batch("hor", [pats[0]] * ((1 << 18) + 1), text, per_pattern_times=False)
    text.free()


def test_kmp_four_bytes_per_step(oracle):
    """kmp_runs on a text of at most four distinct byte values takes four text bytes per table step (kmp_runs<.., FOUR>:
    the rows 4s + 2, which the workgroup derives from the byte table with the text's own two-bit codes): occurrences
    that end anywhere in a dword, at every boundary, dense and overlapping ones, symbols that are not 0..3 (ACGT),
    patterns over fewer symbols than the text, windows beyond 62 bytes (the prefix's automaton + verification), a text
    that ALSO holds a fifth value (the byte-wise kernel on the same plan) — against the oracle, with the table and
    without it (tune(3,5))."""
    from smart_amd import engine
    rng = np.random.default_rng(6)
    n = 3 << 20
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for sigma, letters, m, fifth in ((2, None, 9, False), (2, None, 16, False), (4, None, 33, False), (4, acgt, 62, False), (4, acgt, 63, False),
                                     (2, None, 254, False), (3, None, 300, False), (4, acgt, 2, False), (2, None, 5, False), (4, acgt, 40, True),
                                     (3, None, 17, True)):
        T = oracle.gen_text(777 + m, sigma, 0, n)
        if letters is not None:
            T = letters[T]
        P = T[200_000:200_000 + m].copy()
        if m == 16:
            P[:] = T[200_000]  # one symbol: a pattern over fewer symbols than the text, overlapping occurrences
        for k in rng.integers(0, n - m, 300):            # whole occurrences anywhere, some overlapping
            T[k:k + m] = P
        if fifth:
            T[rng.integers(0, n, 5000)] = 200            # a fifth value: this text is scanned a byte per step
        T[n - m:] = P
        text = Text.upload(T)
        assert text.alphabet() == sorted(set(T.tolist())) and (len(text.alphabet()) > 4 or not fifth or sigma == 3)  # {0, 1, 2, 200}: four values no two bits separate
        want = oracle.search("kmp", P, T)
        assert want >= 100
        engine.tune(0, 1)  # KMP on its own kernel at any length
        try:
            assert smart_amd.kernel_for("kmp", P) == "kmp_runs"
            got_four = smart_amd.search("kmp", P, text)[0]
            engine.tune(3, 5)
            got_plain = smart_amd.search("kmp", P, text)[0]
            engine.tune(3, 0)
            sub = smart_amd.search("kmp", P, text, off=123_457, n=1_500_001)[0]
        finally:
            engine.tune(3, 0)
            engine.tune(0, 0)
        assert got_four == want and got_plain == want, (sigma, m, got_four, got_plain, want)
        assert sub == oracle.search("kmp", P, T[123_457:123_457 + 1_500_001]), (sigma, m)
        text.free()
    # a pattern over FIVE symbols on a four-symbol text (it cannot occur; the byte-wise kernel says so)
    T = acgt[oracle.gen_text(5, 4, 0, n)]
    text = Text.upload(T)
    engine.tune(0, 1)
    try:
        assert smart_amd.search("kmp", np.frombuffer(b"ACGTNACGT", dtype=np.uint8), text)[0] == 0
    finally:
        engine.tune(0, 0)
    text.free()


def test_bndm_gram_tables_on_texts_of_at_most_four_values(oracle):
    """bndm_scan<.., GRAM> (round 4): on a text of at most four distinct byte values an iteration's Q bytes are one of 256
    grams — four two-bit symbols, or eight one-bit symbols on two values — and ONE lookup in a table the workgroup derives
    from the masks and the text's codes replaces the Q mask lookups; a window that is a single gram (8 bytes on two values,
    4 on four) takes bndm.c's occurrence flag and shift from a second table.  Against the oracle and against the mask loop
    (tune(1,9)): byte values that are not 0..3 (ACGT, 0 / 255, one value), planted, overlapping and periodic occurrences,
    windows of one, two and eight grams, lengths that are no whole grams (the mask loop), m > 32 (prefix + verification),
    a fifth value in the text (no gram table), patterns with symbols the text does not hold, sub-ranges, BNDML's m <= 32."""
    from smart_amd import engine
    rng = np.random.default_rng(16)
    n = 2 << 20
    alphabets = {"01": [0, 1], "acgt": list(b"ACGT"), "far": [0, 255], "one": [7], "three": [1, 2, 200], "odd": [3, 5]}
    for name, values in alphabets.items():
        vals = np.asarray(values, dtype=np.uint8)
        for m in (4, 8, 12, 16, 20, 24, 32, 33, 40, 64, 300, 2100, 4200):
            T = vals[oracle.gen_text(991 + m, max(len(values), 2), 0, n) % len(values)]
            P = T[150_000:150_000 + m].copy()
            if m in (8, 24):
                P[:] = np.resize(P[:3], m)           # a period of three: overlapping occurrences, borders
            for k in rng.integers(0, n - m, 200):
                T[k:k + m] = P
            T[:m] = P
            T[n - m:] = P
            if m == 20 and len(values) > 1:
                T[1000:1000 + 3 * m] = np.resize(P, 3 * m)  # back to back
            text = Text.upload(T)
            assert len(text.alphabet()) <= 4
            want = oracle.search("bndm", P, T)
            assert want >= (100 if m <= 300 else 20) or name == "one"  # (long plants overwrite one another)
            sub_want = oracle.search("bf", P, T[54_321:54_321 + 1_000_001])
            engine.tune(0, 1)  # bndm_scan itself at any length
            try:
                for a in ("bndm", "bndml") if m <= 32 else ("bndm",):
                    assert smart_amd.kernel_for(a, P) == "bndm_scan"
                    got = smart_amd.search(a, P, text)[0]
                    engine.tune(1, 9)  # the mask loop on the same plan
                    plain = smart_amd.search(a, P, text)[0]
                    engine.tune(1, 0)
                    assert got == want and plain == want, (name, m, a, got, plain, want)
                    assert smart_amd.search(a, P, text, off=54_321, n=1_000_001)[0] == sub_want, (name, m, a)
                # Horspool on grams (k_horg.hip: the bad-character rule on the window's last gram, the table built by the
                # workgroup from the pattern's last positions) and its byte / hash tables (tune(2,4)) on the same plan
                # (and Boyer-Moore on grams, k_bmg.hip: the same loop with the good-suffix shifts joined in; periodic patterns
                # — m = 8, 24 above — are where bmGs[0] after an occurrence and the shifts of partial matches differ from Horspool's)
                for a in ("hor", "bm", "tunedbm"):
                    got = smart_amd.search(a, P, text)[0]
                    engine.tune(2, 4)
                    plain = smart_amd.search(a, P, text)[0]
                    engine.tune(2, 0)
                    assert got == want and plain == want, (name, m, a, got, plain, want)
                    assert smart_amd.search(a, P, text, off=54_321, n=1_000_001)[0] == sub_want, (name, m, a)
                # a pattern set in one grid (texts up to 32 MiB): the grams of every pattern of the set
                pats = [P, T[77:77 + m].copy(), T[999_999:999_999 + m].copy()]
                pats_want = [oracle.search("bf", p, T) for p in pats]
                for a in ("bndm", "hor", "bm"):
                    counts, _, _, _ = smart_amd.search_batch(a, pats, text)
                    assert counts.tolist() == pats_want, (name, m, a)
                # EPSM on such a text: v_mqsad_pk_u16_u8 at every length (modes 5-9: one to four references of four bytes, the rest
                # of a long pattern verified), its dword compares (tune(7,9)) and the packed-symbol modes of two-value texts
                # (tune(7,5)) on the same plan; tune(7,6): the v_mqsad modes on any text
                assert smart_amd.kernel_for("epsm", P) == "packed_scan"
                got = smart_amd.search("epsm", P, text)[0]
                others = []
                for t in (9, 5, 6):
                    engine.tune(7, t)
                    others.append(smart_amd.search("epsm", P, text)[0])
                engine.tune(7, 0)
                assert got == want and others == [want] * 3, (name, m, "epsm", got, others, want)
                assert smart_amd.search("epsm", P, text, off=54_321, n=1_000_001)[0] == sub_want, (name, m, "epsm")
                # a symbol the text does not hold: no occurrence, whatever the tables say about codes that are not in use
                Q = P.copy()
                Q[m // 2] = 99
                assert smart_amd.search("bndm", Q, text)[0] == 0 and smart_amd.search("epsm", Q, text)[0] == 0 and smart_amd.search("hor", Q, text)[0] == 0 and smart_amd.search("bm", Q, text)[0] == 0, (name, m)
                Q = P.copy()
                Q[m - 1] = 98   # (beyond the sixteen symbols EPSM compares packed, for the longer patterns)
                assert smart_amd.search("bndm", Q, text)[0] == 0 and smart_amd.search("epsm", Q, text)[0] == 0 and smart_amd.search("hor", Q, text)[0] == 0 and smart_amd.search("bm", Q, text)[0] == 0, (name, m)
            finally:
                engine.tune(1, 0)
                engine.tune(2, 0)
                engine.tune(7, 0)
                engine.tune(0, 0)
            text.free()
    # a fifth value: the text has no codes, the mask loop runs
    T = np.asarray(list(b"ACGT"), dtype=np.uint8)[oracle.gen_text(31, 4, 0, n)]
    T[rng.integers(0, n, 1000)] = ord("N")
    P = T[5000:5016].copy()
    text = Text.upload(T)
    engine.tune(0, 1)
    try:
        assert smart_amd.search("bndm", P, text)[0] == oracle.search("bf", P, T)
    finally:
        engine.tune(0, 0)
    text.free()


def test_occurrence_sums_of_large_grids_are_staged_and_cleared(oracle):
    """flush_hits (dev_common.hpp): a grid of 1024+ workgroups adds its workgroups' sums to 64 staging slots in the text's
    front pad, the last arrival of a slot forwards it and clears it.  Occurrences in EVERY workgroup (two values, m = 2:
    a quarter of all positions), the same text searched again and again by kernels of every family, whole and in part,
    one pattern and a set in one grid (which adds directly): every count equals the oracle's on a slice and the
    first search's on the whole — nothing left behind in a slot, nothing counted twice."""
    from smart_amd import engine
    n = 300_000_000
    text = Text.generate(SEED2, 2, n)
    T = oracle.gen_text(SEED2, 2, 0, 4_000_000)
    assert bytes(text.read(0, 4096)) == T[:4096].tobytes()
    for m in (2, 3, 9):
        P = T[1000:1000 + m].copy()
        small = oracle.search("bf", P, T)
        engine.tune(0, 1)  # every algorithm on its own kernel: tiles (hor, bm, bndm), runs (so, kmp), packed (epsm)
        try:
            whole = {}
            for rep in range(3):
                for a in ("epsm", "hor", "bm", "bndm", "so", "kmp"):
                    c = smart_amd.search(a, P, text)[0]
                    assert whole.setdefault(m, c) == c, (a, m, rep, c, whole)
                    assert smart_amd.search(a, P, text, off=0, n=4_000_000)[0] == small, (a, m, rep)
            pats = [P, T[5000:5000 + m].copy(), T[77:77 + m].copy()]
            counts, _, _, _ = smart_amd.search_batch("epsm", pats, text)
            assert counts[0] == whole[m] and all(smart_amd.search("so", p, text)[0] == c for p, c in zip(pats, counts.tolist())), (m, counts)
        finally:
            engine.tune(0, 0)
    text.free()


def test_text_alphabet():
    """What a text consists of is taken on the device when it is created (smartgpu_text_alphabet): uploaded, tiled and
    generated texts, lengths that are no multiple of the 16-byte loads, the empty text."""
    rng = np.random.default_rng(11)
    for n, values in ((0, [0]), (1, [9]), (15, [1, 2]), (16, [255]), (17, [0, 255]), (100_003, [65, 67, 71, 84]),
                      (1 << 20, list(range(256))), ((1 << 20) + 5, [3, 77, 200, 201, 202])):
        T = np.asarray(values, dtype=np.uint8)[rng.integers(0, len(values), n)]
        if n > 16:
            T[-1] = values[-1]  # a value that may occur in the last, partial 16 bytes only
            T[:-1][T[:-1] == values[-1]] = values[0]
        text = Text.upload(T)
        assert text.alphabet() == sorted(set(T.tolist())), (n, values)
        text.free()
    text = Text.generate(SEED2, 4, 1 << 20)
    assert text.alphabet() == [0, 1, 2, 3]
    text.free()
    unit = np.frombuffer(b"abracadabra", dtype=np.uint8)
    text = Text.upload_tiled(unit, 100_000)
    assert text.alphabet() == sorted(set(unit.tolist()))
    text.free()


def test_four_symbol_texts_on_shift_or_runs(oracle):
    """On a text of at most four distinct byte values so_runs takes four bytes per table step (so_runs<., FOUR>: the
    text's two-bit codes index a table of ready-made four-step operands).  Alphabets that are not 0..3 (ACGT: codes from
    bits 1-2), two values far apart, one value, three values, four values that no pair of adjacent bits separates (the
    byte-wise kernel then); patterns with symbols the text does not hold; 30+ byte patterns (prefix + verification);
    sub-ranges; Shift-And's complemented masks; other algorithms' short patterns, which count on so_runs — against the
    oracle, with the table and without (tune(6,5))."""
    from smart_amd import engine
    rng = np.random.default_rng(12)
    n = (3 << 20) + 77
    cases = (("0123", [0, 1, 2, 3], True), ("ACGT", list(b"ACGT"), True), ("far", [7, 200], True), ("one", [42], True),
             ("three", [10, 11, 13], True), ("unseparable", [0, 1, 2, 4], False), ("five", [0, 1, 2, 3, 4], False))
    for name, values, four in cases:
        letters = np.asarray(values, dtype=np.uint8)
        T = letters[rng.integers(0, len(values), n)]
        text = Text.upload(T)
        assert text.alphabet() == sorted(values)
        for m in (1, 2, 3, 4, 5, 8, 12, 16, 29, 30, 33, 64, 300):
            k = int(rng.integers(0, n - m))
            P = T[k:k + m].copy()
            pats = [P]
            if m >= 2:
                Q = P.copy()
                Q[m // 2] = 250  # a symbol the text does not hold
                pats.append(Q)
            if m >= 8:  # planted, overlapping copies of a periodic pattern (not in a shared text: a copy)
                pats.append(np.resize(letters[rng.integers(0, len(values), 3)], m))
            for P in pats:
                want = oracle.search("bf", P, T)
                sub_want = oracle.search("bf", P, T[100_001:100_001 + 2_000_003])
                for algo in ("so", "sa", "hor", "bndm", "epsm", "kmp"):
                    if smart_amd.kernel_for(algo, P) != "so_runs":
                        continue
                    got = smart_amd.search(algo, P, text)[0]
                    sub = smart_amd.search(algo, P, text, off=100_001, n=2_000_003)[0]
                    engine.tune(6, 5)
                    try:
                        plain = smart_amd.search(algo, P, text)[0]
                    finally:
                        engine.tune(6, 0)
                    assert got == want and plain == want and sub == sub_want, (name, algo, m, got, plain, want, sub, sub_want)
        text.free()
    # dense periodic matches across run boundaries
    T = np.resize(np.frombuffer(b"ACACACGT", dtype=np.uint8), n).copy()
    text = Text.upload(T)
    for P in (b"ACAC", b"CACACGTACACAC", b"ACACACGT" * 5):
        P = np.frombuffer(P, dtype=np.uint8)
        assert smart_amd.search("so", P, text)[0] == oracle.search("bf", P, T)
    text.free()


def test_long_patterns_with_many_candidates_per_lane_on_the_skip_kernels(oracle):
    """The long-pattern instantiations of bm_scan, hor_scan's flat form and bndm_scan count the candidates of a lane's
    segment in the walk and verify once per tile; a lane that saw MORE than one walks its segment again and compares on the
    spot.  Periodic texts and patterns put several candidates — occurrences and near-occurrences — into every 64-byte
    segment: every algorithm on its own kernel (tune(0,1)), whole text and a sub-range, against the oracle."""
    from smart_amd import engine
    rng = np.random.default_rng(77)
    n = 200_000 + 13
    texts = {"a": np.full(n, ord("a"), dtype=np.uint8),
             "ab": np.resize(np.frombuffer(b"ab", dtype=np.uint8), n),
             "abc+noise": np.resize(np.frombuffer(b"abcabcabd", dtype=np.uint8), n).copy()}
    texts["abc+noise"][rng.integers(0, n, 400)] = ord("x")      # near-occurrences that fail in the part verified from memory
    english = np.frombuffer(open(os.path.join(GOLDEN, "english_excerpt.txt"), "rb").read(), dtype=np.uint8)[:n].copy()
    texts["english+copies"] = english
    engine.tune(0, 1)
    try:
        for name, T in texts.items():
            text = Text.upload(T)
            for m in (18, 19, 33, 40, 64, 65, 300, 1000):
                if name == "english+copies":
                    P = T[5000:5000 + m].copy()
                    for k in range(20_000, 20_000 + 40 * (m + 3), m + 3):  # copies a few bytes apart, and broken ones
                        T2 = P.copy()
                        if (k // (m + 3)) % 3 == 0:
                            T2[0] ^= 1  # fails at the FIRST byte: the last thing the skip loops compare
                        T[k:k + m] = T2
                    text.free()
                    text = Text.upload(T)
                else:
                    P = T[7:7 + m].copy()
                want = oracle.search("bf", P, T)
                sub_want = oracle.search("bf", P, T[1234:1234 + 150_001])
                for algo in ("hor", "bm", "bndm", "tunedbm", "bndml"):
                    got = smart_amd.search(algo, P, text)[0]
                    sub = smart_amd.search(algo, P, text, off=1234, n=150_001)[0]
                    assert got == want and sub == sub_want, (name, algo, m, got, want, sub, sub_want)
            text.free()
    finally:
        engine.tune(0, 0)
