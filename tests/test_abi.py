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


def test_compute_fails_loudly_without_gpu():
    """No CPU fallback: on a host without a HIP device the search entry points
    return an error (on a GPU box this test is a no-op)."""
    if smart_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    T = np.zeros(64, dtype=np.uint8)
    assert smart_amd.search_host("hor", T[:4], T) == -1
    with pytest.raises(smart_amd.SmartGpuError):
        smart_amd.Text.upload(T)
