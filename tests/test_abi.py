"""CPU tests of the drop-in boundary: the library loads, exports every symbol
include/smartgpu.h declares, and its host-side preprocessing equals the oracle's
(no compute calls — those need a GPU and live in test_parity_gpu.py)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

import smart_amd
from smart_amd import engine


@pytest.fixture(scope="module", autouse=True)
def built():
    engine.build()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "smartgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smartgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(engine.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(L, n), n


def test_algorithm_registry():
    assert smart_amd.version().startswith("smartgpu")
    for i, a in enumerate(engine.ALGOS):
        assert smart_amd.algo_id(a) == i
        assert smart_amd.algo_id(a.upper()) == i  # smart.c:142 lower-cases names
        assert engine.lib().smartgpu_algo_name(i).decode() == a
    with pytest.raises(smart_amd.SmartGpuError):
        smart_amd.algo_id("nope")


def test_tables_match_oracle(oracle):
    rng = np.random.default_rng(7)
    pats = [b"a", b"aa", b"ab" * 20, b"abcabcabd", b"a" * 40, b"gcagagag"]
    for sigma in (2, 4, 128, 256):
        for m in (1, 2, 3, 8, 31, 32, 33, 64, 255, 256, 257, 1000, 4200):
            pats.append(oracle.gen_text(int(rng.integers(0, 2**40)), sigma, 0, m).tobytes())
    for p in pats:
        P = np.frombuffer(p, dtype=np.uint8)
        assert np.array_equal(smart_amd.build_table("bad_char", P), oracle.tables("hor", P))
        assert np.array_equal(smart_amd.build_table("good_suffix", P), oracle.tables("bm_gs", P))
        assert np.array_equal(smart_amd.build_table("kmp_next", P), oracle.tables("kmp", P))
        w = min(len(P), 32)
        S, lim = oracle.tables("so", P[:w])
        assert np.array_equal(smart_amd.build_table("shift_or", P).view(np.uint32), S)
        # so.c:56 tests D < lim; the kernel tests bit w-1 — same predicate on any D with bits >= w set
        assert lim == (~(((1 << w) - 1) >> 1)) & 0xFFFFFFFF
        assert np.array_equal(smart_amd.build_table("bndm", P).view(np.uint32), oracle.tables("bndm", P[:w]))
        # sa.c:27-34: the complement of so.c's masks over the first w bytes
        assert np.array_equal(smart_amd.build_table("shift_and", P).view(np.uint32), (~S) & np.uint32((1 << w) - 1 if w < 32 else 0xFFFFFFFF))
        # qs.c:27-31: m+1 for a byte that does not occur, else m - (right-most position)
        qs = np.full(256, len(P) + 1, dtype=np.int32)
        for i, c in enumerate(P):
            qs[c] = len(P) - i
        assert np.array_equal(smart_amd.build_table("quick_search", P), qs)
        # hash3.c:36-56 (and hash5.c, hash8.c): shifts under the 8-bit q-gram hash, then the shift after a candidate
        for q in (3, 5, 8):
            if len(P) < q:
                continue
            m = len(P)
            h = lambda end: sum(int(P[end - k]) << k for k in range(q)) & 0xFF  # noqa: E731
            sh = np.full(256, m - q + 1, dtype=np.int32)
            for i in range(q - 1, m - 1):
                sh[h(i)] = m - 1 - i
            after = max(int(sh[h(m - 1)]), 1)
            sh[h(m - 1)] = 0
            got = smart_amd.build_table("hash%d" % q, P)
            assert np.array_equal(got[:256], sh) and got[256] == after, (q, m)


def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: on a host without a HIP device the search entry points
    return an error (on a GPU box this test is a no-op)."""
    if smart_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    T = np.zeros(64, dtype=np.uint8)
    assert smart_amd.search_host("hor", T[:4], T) == -1
    with pytest.raises(smart_amd.SmartGpuError):
        smart_amd.Text.upload(T)


def test_kmp_transition_tables_count_like_the_oracle(oracle):
    """The KMP kernels run the failure function expanded into delta[state][byte]
    (full and own-alphabet forms); walking those tables on the host must count
    exactly what kmp.c counts."""
    rng = np.random.default_rng(11)
    for sigma, n, m in ((2, 3000, 5), (2, 3000, 40), (4, 4000, 17), (128, 5000, 3), (250, 5000, 200), (3, 2000, 255)):
        T = oracle.gen_text(int(rng.integers(0, 2**40)), sigma, 0, n)
        for P in (T[100:100 + m].copy(), np.resize(T[7:9], m)):
            want = oracle.search("kmp", P, T)
            dfa = smart_amd.build_table("kmp_dfa", P).reshape(m + 1, 256)
            comp = smart_amd.build_table("kmp_dfa_compressed", P)
            k1, colmap, table = int(comp[0]), comp[1:257], comp[257:].reshape(m + 1, -1)
            assert table.shape[1] == k1
            st = st2 = hits = hits2 = 0
            for c in T.tolist():
                st = int(dfa[st, c])
                hits += st == m
                st2 = int(table[st2, colmap[c]])
                hits2 += st2 == m
            assert hits == want and hits2 == want, (sigma, n, m, hits, hits2, want)


def test_host_table_builders_under_sanitizers(tmp_path):
    """tables.cpp (the host-side preprocessing the kernels depend on) built with
    AddressSanitizer + UBSan on the CPU and driven over many patterns."""
    import subprocess
    src = os.path.join(ROOT, "smart_amd", "csrc", "tables.cpp")
    drv = tmp_path / "drv.cpp"
    drv.write_text(r"""
#include "tables.hpp"
#include <cstdio>
#include <vector>
int main() {
    unsigned long long x = 88172645463325252ull, sink = 0;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    const unsigned ms[] = {1, 2, 3, 4, 7, 8, 31, 32, 33, 64, 65, 95, 96, 255, 256, 1000, 4200};
    for (unsigned sigma : {1u, 2u, 4u, 128u, 256u})
        for (unsigned m : ms)
            for (int rep = 0; rep < 3; ++rep) {
                std::vector<uint8_t> P(m);
                for (auto& b : P) b = (uint8_t)(rnd() % sigma);
                sink += sg::bad_char(P.data(), m)[P[0]];
                sink += sg::good_suffix(P.data(), m)[0];
                sink += sg::kmp_next(P.data(), m)[m];
                sink += sg::shift_or_masks(P.data(), m)[P[0]];
                sink += sg::bndm_masks(P.data(), m)[P[0]];
                sink += sg::shift_and_masks(P.data(), m)[P[0]];
                sink += sg::quick_search_shifts(P.data(), m)[P[0]];
                for (unsigned q : {3u, 5u, 8u})
                    if (m >= q) {
                        int32_t after = 0;
                        sink += sg::qgram_hash_shifts(P.data(), m, q, &after)[P[m - 1]] + after;
                    }
                {
                    std::vector<uint8_t> t(m % 7, 0);  // appended after what the blob already holds
                    sg::kmp_runs_tables(P.data(), m < 254 ? m : 254, t);
                    sink += t.back() + t.size();
                }
                if (m <= 255) {
                    uint32_t k1 = 0;
                    sink += sg::kmp_dfa(P.data(), m)[m * 256u + P[0]];
                    sink += sg::kmp_dfa_compressed(P.data(), m, &k1).size() + k1;
                }
            }
    std::printf("ok %llu\n", sink);
    return 0;
}
""")
    exe = tmp_path / "drv"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "smart_amd", "csrc"), str(drv), src, "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_kmp_runs_tables_are_the_automaton_renumbered(oracle):
    """kmp_runs' tables (tables.cpp kmp_runs_tables, built row by row in place) against the layout DESIGN.md §4
    states, reconstructed here from the plain transition table: row id(s) XOR-swizzled by its id, transitions into
    the accept state lead to the absorbing row Z; Q[s] = P[s..s+4) for the borderless states 0..K, thr = 4K."""
    rng = np.random.default_rng(5)
    cases = [oracle.gen_text(77 + i, sigma, 0, m) for i, (sigma, m) in enumerate(
        [(2, 5), (2, 9), (2, 40), (2, 62), (2, 63), (2, 64), (2, 200), (2, 254), (4, 17), (4, 100), (4, 254), (128, 1), (128, 2),
         (128, 4), (128, 5), (128, 6), (128, 32), (128, 62), (128, 63), (128, 191), (128, 192), (128, 193), (128, 254), (128, 300)])]
    cases += [np.frombuffer(b"abcabcabcabd", np.uint8), np.frombuffer(b"aaaaaaaaab", np.uint8), np.frombuffer(b"abcdeabcdeabcdf", np.uint8)]
    cases += [np.tile(np.frombuffer(b"ab", np.uint8), 120), np.concatenate([rng.integers(0, 256, 250, dtype=np.uint8), [7, 7, 7, 7]]).astype(np.uint8)]
    for P in cases:
        w = len(P) if len(P) <= 254 else 62  # kernels.hpp kmp_window
        Pw = P[:w]
        dfa = smart_amd.build_table("kmp_dfa", Pw).reshape(w + 1, 256)
        got = smart_amd.build_table("kmp_runs", P).astype(np.uint8)
        small = w < 63
        idw = 4 * w if small else 254
        Z = idw + 1
        rot = lambda s: ((s << 2) | (s >> 6)) & 255  # noqa: E731
        ident = lambda s: idw if s == w else 4 * s if small else rot(w) if s == 191 else rot(s)  # noqa: E731
        assert len(got) == (Z + 1) * 256 + 272, (len(P), len(got))
        want = np.zeros((Z + 1, 256), np.uint8)
        ids = [ident(s) for s in range(w + 1)]
        assert len(set(ids)) == w + 1 and Z not in ids
        for s in range(w + 1):
            r = ids[s]
            for c in range(256):
                nx = int(dfa[s, c])
                want[r, c ^ r] = Z if nx == w else ids[nx]
        want[Z, :] = Z
        assert np.array_equal(got[:(Z + 1) * 256].reshape(Z + 1, 256), want), len(P)
        q = got[(Z + 1) * 256:].view(np.uint32)
        border = [0, 0]
        for s in range(2, w + 1):  # longest proper border of P[0..s)
            k = border[s - 1]
            while k and Pw[s - 1] != Pw[k]:
                k = border[k]
            border.append(k + 1 if Pw[s - 1] == Pw[k] else 0)
        K = 0
        if w >= 5:
            while K < min(w - 5, 58) and border[K + 1] == 0:
                K += 1
            for s in range(K + 1):
                assert q[s] == int.from_bytes(bytes(Pw[s:s + 4]), "little"), (len(P), s)
        assert q[64] == 4 * K and not q[65:].any() and not q[K + 1:64].any()


def test_kmp_compact_tables_are_the_automaton_renumbered(oracle):
    """The COMPACT tables of kmp_runs<., false, COMPACT> (round 4: five four-wave workgroups per CU, each with its own
    table): the automaton of the pattern or of its 56-byte prefix, state s in row s (id 4s, the row XOR-swizzled by its
    id), transitions into the accept state lead to the absorbing row Z = 4 (w + 1) that follows the states' rows."""
    cases = [oracle.gen_text(177 + i, sigma, 0, m) for i, (sigma, m) in enumerate(
        [(2, 1), (2, 5), (2, 9), (2, 40), (2, 55), (2, 56), (2, 57), (2, 200), (4, 17), (4, 100), (128, 1), (128, 2), (128, 4),
         (128, 5), (128, 32), (128, 56), (128, 57), (128, 62), (128, 254), (128, 255), (128, 4096), (256, 33)])]
    cases += [np.frombuffer(b"abcabcabcabd", np.uint8), np.frombuffer(b"aaaaaaaaab", np.uint8), np.tile(np.frombuffer(b"ab", np.uint8), 120)]
    for P in cases:
        w = min(len(P), 56)  # kernels.hpp kmp_compact_window
        Pw = P[:w]
        dfa = smart_amd.build_table("kmp_dfa", Pw).reshape(w + 1, 256)
        got = smart_amd.build_table("kmp_runs_compact", P).astype(np.uint8)
        Z = 4 * w + 4
        assert Z <= 252 and len(got) == (w + 2) * 256 + 272, (len(P), len(got))
        want = np.zeros((w + 2, 256), np.uint8)
        for s in range(w + 1):
            for c in range(256):
                nx = int(dfa[s, c])
                want[s, c ^ (4 * s)] = Z if nx == w else 4 * nx
        want[w + 1, :] = Z
        assert np.array_equal(got[:(w + 2) * 256].reshape(w + 2, 256), want), len(P)
        # Q and thr are the spread tables' (same states, same ids)
        if len(P) <= 56:
            spread = smart_amd.build_table("kmp_runs", P).astype(np.uint8)
            assert np.array_equal(got[(w + 2) * 256:], spread[-272:])


def test_multi_gpu_partition_arithmetic_of_the_c_library():
    """smartgpu_mtext_partition — what smartgpu_mtext_upload / _generate shard a text with (api.cpp shard_begin): for k up to
    16 and lengths that are not multiples of k the shards' own ranges tile [0, n) exactly, differ by at most one byte, hold
    SMARTGPU_XSIZE - 1 bytes of overlap (never beyond n) — and counting starts by ownership gives every start position of
    every pattern length to exactly one shard."""
    X = 4200
    for n in (0, 1, 7, 8, 9, 1_000_003, (1 << 32) + 5, (1 << 35) - 1, (1 << 35) + 12345):
        for k in (1, 2, 3, 5, 8, 16):
            parts = [engine.mtext_partition(n, k, g) for g in range(k)]
            assert parts[0][0] == 0 and sum(o for _, o, _ in parts) == n
            for g, (b, o, h) in enumerate(parts):
                assert b == sum(x[1] for x in parts[:g])                     # contiguous, in order
                assert o in (n // k, n // k + 1)                            # balanced
                assert h == min(n, b + o + X - 1) - b and b + h <= n        # own + overlap, inside the text
            assert [o for _, o, _ in parts] == sorted((o for _, o, _ in parts), reverse=True)  # the longer shards first
            # ownership by start position: shard g counts starts s in [b, b+o) with s + m <= n — smartgpu_msearch* searches
            # span = min(held, own + m - 1) bytes of the shard, i.e. starts [b, b + span - m + 1)
            for m in (1, 2, 4200):
                owned = 0
                for b, o, h in parts:
                    span = min(h, o + m - 1)
                    owned += max(0, span - m + 1)
                assert owned == max(0, n - m + 1), (n, k, m)
    with pytest.raises(smart_amd.SmartGpuError):
        engine.mtext_partition(100, 8, 8)
    with pytest.raises(smart_amd.SmartGpuError):
        engine.mtext_partition(100, 17, 0)


def test_launch_pool_of_the_multi_gpu_search():
    """The host threads that enqueue the k devices' launches of smartgpu_msearch* at once (api.cpp LaunchPool), without
    a device: every job of every round runs exactly once, with the number of jobs changing from round to round."""
    L = engine.lib()
    for k in (1, 2, 8, 16):
        assert L.smartgpu_selftest_launch_pool(k, 200) == 0, L.smartgpu_last_error().decode()
    assert L.smartgpu_selftest_launch_pool(0, 1) != 0 and L.smartgpu_selftest_launch_pool(17, 1) != 0


def test_kernel_choice_follows_the_pattern(oracle):
    """The plan picks the kernel from the pattern's own symbols (api.cpp build_blob);
    smartgpu_kernel_for answers without a device."""
    from smart_amd import engine
    kf = smart_amd.kernel_for
    rnd = oracle.gen_text(123, 128, 0, 5000)       # rand128: symbols do not repeat
    eng = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "english_excerpt.txt"), "rb").read(5000), dtype=np.uint8)
    two = oracle.gen_text(9, 2, 0, 5000)
    four = oracle.gen_text(9, 4, 0, 5000)
    # own LDS-tile skip kernels on random text over a large alphabet
    for m in (32, 64, 256, 4096):
        assert kf("hor", rnd[:m]) == "hor_scan" and kf("bm", rnd[:m]) == "bm_scan" and kf("bndm", rnd[:m]) == "bndm_scan"
        assert kf("bndml", rnd[:m]) == ("bndm_scan" if m <= 32 else "bndml_scan")
        assert kf("kr", rnd[:m]) == "hor_scan_bp"
    # short patterns (crossovers per algorithm): the Shift-Or runs kernel (round 1: the packed matcher)
    assert kf("hor", rnd[:7]) == "so_runs" and kf("hor", rnd[100:108]) in ("hor_scan", "so_runs")
    assert kf("bm", rnd[:7]) == "so_runs" and kf("bndm", rnd[:10]) == "so_runs" and kf("kr", rnd[:15]) == "so_runs" and kf("sbndm", rnd[:10]) == "so_runs"
    # natural language, DNA-like alphabets: symbols repeat -> the Shift-Or runs kernel at any m (round 1: packed matcher) —
    # except, since round 3, where the algorithm's own kernel holds on such patterns: the flat loops of bm_scan and
    # hor_scan on natural language from 8 bytes on (not on a few symbols), bndm_scan with q-grams from 16 bytes on (two
    # symbols: from 32); since round 4 Horspool on grams (a text of at most four byte values; on any other text its q-gram
    # hash table) for patterns over two to four symbols from 16 bytes on, BNDM's gram form from 8
    for m in (16, 64, 1024):
        for a in ("hor", "bm", "bndm", "qs", "raita", "hash3", "sbndm", "bndml", "tunedbm"):
            own_eng = {"bm": "bm_scan", "hor": "hor_scan", "tunedbm": "hor_scan", "bndm": "bndm_scan", "bndml": "bndm_scan" if m <= 32 else "so_runs"}.get(a, "so_runs")
            own_four = {"bndm": "bndm_scan", "bndml": "bndm_scan" if m <= 32 else "so_runs", "hor": "hor_scan", "tunedbm": "hor_scan", "bm": "bm_scan"}.get(a, "so_runs")  # (round 4: Horspool, Tuned BM and Boyer-Moore on grams from 16 bytes on)
            assert kf(a, eng[200:200 + m]) == own_eng, (a, m)
            assert kf(a, four[:m]) == own_four, (a, m)
    assert kf("bm", eng[200:207]) == "so_runs" and kf("bm", eng[200:208]) == "bm_scan" and kf("bndm", eng[200:212]) == "so_runs" and kf("bndm", four[:12]) == "bndm_scan"  # (four symbols, 8+ bytes: the gram form)
    assert kf("hor", b"abca") == "so_runs" and kf("bm", four[:4]) == "so_runs" and kf("hor", b"abcd") == "so_runs"
    # two symbols, 16+ bytes: the bit-parallel runs kernel, whatever the algorithm (KMP and KR keep their own, Horspool and Boyer-Moore their gram forms from 32 bytes on, EPSM is
    # the packed matcher except on patterns its first dword cannot tell apart; BNDM its own from 32 bytes on)
    for m in (16, 33, 300):
        for a in engine.ALGOS:
            want = {"kmp": "kmp_runs", "kr": "hor_scan_bp", "bndm": "bndm_scan", "bndml": "bndm_scan" if m <= 32 else "so_runs",
                    "hor": "hor_scan" if m >= 32 else "so_runs", "tunedbm": "hor_scan" if m >= 32 else "so_runs", "bm": "bm_scan" if m >= 32 else "so_runs"}.get(a, "so_runs")
            assert kf(a, two[:m]) == want, (a, m)
    # (EPSM, round 4: its v_mqsad references decide up to 12 bytes on two symbols at the runs kernel's pace: it keeps those)
    assert kf("hor", two[:32]) == "hor_scan" and kf("bm", two[:32]) == "bm_scan" and kf("hor", two[:31]) == "so_runs" and kf("hor", four[:16]) == "hor_scan" and kf("hor", four[:15]) == "so_runs" and kf("epsm", two[:12]) == "packed_scan" and kf("epsm", two[:13]) == "so_runs" and kf("epsm", two[:7]) == "packed_scan"
    # round 4: 8+ bytes over two to four symbols: bndm_scan's gram form (one lookup per window on a text of <= 4 byte values)
    assert len(set(two[:8].tolist())) == 2 and kf("bndm", two[:8]) == "bndm_scan" and kf("bndml", two[:8]) == "bndm_scan"
    assert 3 <= len(set(four[:8].tolist())) <= 4 and kf("bndm", four[:8]) == "bndm_scan" and kf("bndm", four[:9]) == "bndm_scan"
    assert kf("bndm", rnd[:8]) == "so_runs" and kf("bndm", four[:7]) == "so_runs" and kf("sbndm", two[:8]) == "so_runs"
    # ... on four symbols EPSM stays the packed matcher at every length (round 4: two v_mqsad references, survivors completed: 0.78)
    assert kf("epsm", four[:8]) == "packed_scan" and kf("epsm", four[:64]) == "packed_scan" and kf("epsm", four[:4]) == "packed_scan"
    eight = oracle.gen_text(9, 8, 0, 5000)
    assert kf("epsm", eight[:64]) == "packed_scan" and kf("epsm", rnd[:8]) == "packed_scan"
    # the serial automata never move — KMP from 9 bytes on (below that its automaton has no room for the 0..K form)
    for P in (rnd[:32], eng[:32], four[:32], rnd[:9], two[:9]):
        assert kf("kmp", P) == "kmp_runs" and kf("so", P) == "so_runs" and kf("sa", P) == "so_runs"
    assert kf("kmp", rnd[:8]) == "so_runs" and kf("kmp", eng[:2]) == "so_runs" and kf("kmp", rnd[:1]) == "so_runs"
    assert kf("epsm", rnd[:32]) == "packed_scan" and kf("epsm", eng[:32]) == "packed_scan"  # EPSM is the packed matcher
    # tune(0,1): every algorithm on its own kernel
    engine.tune(0, 1)
    try:
        assert kf("bm", eng[200:264]) == "bm_scan" and kf("hor", two[:64]) == "hor_scan" and kf("kr", rnd[:8]) == "hor_scan_bp"
        assert kf("kmp", rnd[:4]) == "kmp_runs"
    finally:
        engine.tune(0, 0)
    with pytest.raises(smart_amd.SmartGpuError):
        kf("hash8", rnd[:7])


def test_product_library_carries_no_superseded_kernels():
    """The product library exports every declared symbol and refuses the smartgpu_tune settings whose kernels
    live only in the A/B build (kernels_ab.inc); the A/B build exports the same ABI and accepts them."""
    from smart_amd import engine
    for key, value in ((3, 1), (3, 2), (3, 3), (6, 1), (6, 2), (6, 3), (6, 4), (7, 1), (7, 3), (0, 2)):
        with pytest.raises(smart_amd.SmartGpuError) as e:
            engine.tune(key, value)
        assert "A/B build" in str(e.value)
    for key, value in ((0, 1), (0, 3), (0, 0), (4, 6), (4, 0), (5, 1024), (5, 0)):
        engine.tune(key, value)
    prod = open(engine.LIB_PATH, "rb").read()
    for name in (b"so_runs64", b"kmp_links_runs", b"so_scan", b"kmp_scan", b"kmp_runs1", b"so_runs1"):
        assert name not in prod, name
    ab = ctypes.CDLL(engine.AB_LIB_PATH)
    for n in declared_symbols():
        assert hasattr(ab, n), n
    assert os.path.getsize(engine.LIB_PATH) < 0.7 * os.path.getsize(engine.AB_LIB_PATH)
    L = engine.use_library(engine.AB_LIB_PATH)
    try:
        assert b"A/B" in L.smartgpu_version()
        engine.tune(3, 3)
        assert smart_amd.kernel_for("kmp", b"abcdabcdabcd") == "kmp_runs1"
        engine.tune(3, 0)
        engine.tune(6, 2)
        assert smart_amd.kernel_for("so", b"abcdabcd") == "so_runs64"
        engine.tune(6, 0)
    finally:
        engine.use_library()
    assert smart_amd.kernel_for("kmp", b"abcdabcdabcd") == "kmp_runs"


def test_four_symbol_codes():
    """The two-bit codes the runs kernels use on a text of at most four distinct byte values (tables.cpp
    four_symbol_codes; build_table "four_codes" takes the set of P's bytes): the lowest shift whose two bits tell the
    members apart, symtab = the member of each code — for a code without a member a byte with that code that is NOT a
    member; a fifth value, or values whose codes collide at every shift: none."""
    for values in ([0, 1, 2, 3], [0, 1], [1], [65, 67, 71, 84], [7, 200], [10, 11, 13], [0x10, 0x20, 0x30], [255, 254, 253, 252], [0, 4, 8, 12], [0, 64, 128, 192]):
        got = smart_amd.build_table("four_codes", np.array(values * 3, dtype=np.uint8))
        assert len(got) == 2, values
        shift, symtab = int(got[0]), int(got[1]) & 0xFFFFFFFF
        codes = {v: (v >> shift) & 3 for v in values}
        assert 0 <= shift < 7 and len(set(codes.values())) == len(values), (values, shift)
        assert all(len({(v >> sh) & 3 for v in values}) < len(values) for sh in range(shift)), (values, shift)  # the lowest
        for c in range(4):
            b = (symtab >> (8 * c)) & 0xFF
            assert (b >> shift) & 3 == c
            assert (b in values) == (c in codes.values()) and (b not in values or codes[b] == c), (values, c, b)
    assert len(smart_amd.build_table("four_codes", np.array([1, 2, 3, 4, 5], dtype=np.uint8))) == 0
    assert len(smart_amd.build_table("four_codes", np.array([0x00, 0x04, 0x10, 0x40], dtype=np.uint8))) == 0  # any two bits tell at most three apart
    assert len(smart_amd.build_table("four_codes", np.array([0, 1, 2, 4], dtype=np.uint8))) == 0
