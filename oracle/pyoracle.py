"""ctypes loader for oracle/libsmartoracle.so (this project's CPU restatement)
and for oracle/_ref/lib<algo>.so (the real reference algorithms, when built).

TEST INFRASTRUCTURE ONLY — see oracle/smart_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsmartoracle.so")
REF_DIR = os.path.join(HERE, "_ref")
ALGOS = ("bf", "hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita", "hash3", "hash5", "hash8", "sbndm", "kr", "bndml")

_lib = None


def build(ref=None):
    """Compile the restatement (always) and the reference builds (when the
    reference tree is present, or when ref=True)."""
    targets = ["all"]
    if ref is None:
        ref = os.path.isdir("/root/reference/src/algos")
    if ref:
        targets += ["ref", "refbin"]
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        u8p = C.c_void_p
        for a in ALGOS:
            f = getattr(L, "oracle_" + a)
            f.restype = C.c_uint64
            f.argtypes = [u8p, C.c_int, u8p, C.c_uint64]
        L.oracle_search.restype = C.c_uint64
        L.oracle_search.argtypes = [C.c_char_p, u8p, C.c_int, u8p, C.c_uint64]
        L.oracle_search_int.restype = C.c_int
        L.oracle_search_int.argtypes = [C.c_char_p, u8p, C.c_int, u8p, C.c_int]
        L.oracle_search_mt.restype = C.c_uint64
        L.oracle_search_mt.argtypes = [C.c_char_p, u8p, C.c_int, u8p, C.c_uint64, C.c_int]
        L.oracle_textgen.restype = C.c_int
        L.oracle_textgen.argtypes = [C.c_int, u8p, C.c_uint64]
        L.oracle_gen_text.restype = None
        L.oracle_gen_text.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, u8p]
        L.oracle_splitmix64.restype = C.c_uint64
        L.oracle_splitmix64.argtypes = [C.c_uint64]
        L.oracle_pre_hor.argtypes = [u8p, C.c_int, u8p]
        L.oracle_pre_bm_gs.argtypes = [u8p, C.c_int, u8p]
        L.oracle_pre_bm_suffixes.argtypes = [u8p, C.c_int, u8p]
        L.oracle_pre_kmp.argtypes = [u8p, C.c_int, u8p]
        L.oracle_pre_so.argtypes = [u8p, C.c_int, u8p]
        L.oracle_pre_so.restype = C.c_uint32
        L.oracle_pre_bndm.argtypes = [u8p, C.c_int, u8p]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    return a


def search(algo, P, T, threads=1):
    """Occurrence count of P in T by this project's restatement of `algo`."""
    P = _u8(P)
    T = _u8(T)
    L = lib()
    if threads > 1:
        return int(L.oracle_search_mt(algo.encode(), P.ctypes.data, len(P), T.ctypes.data, len(T), threads))
    return int(L.oracle_search(algo.encode(), P.ctypes.data, len(P), T.ctypes.data, len(T)))


def textgen(sigma, n):
    """First n bytes of SMART's rand<sigma> corpus (src/textgen.c:34-54)."""
    out = np.empty(n, dtype=np.uint8)
    if lib().oracle_textgen(sigma, out.ctypes.data, n) != 0:
        raise ValueError("textgen does not emit sigma=%d" % sigma)
    return out


def gen_text(seed, sigma, off, n):
    """Bytes off..off+n of the counter-based corpus."""
    out = np.empty(n, dtype=np.uint8)
    lib().oracle_gen_text(seed, sigma, off, n, out.ctypes.data)
    return out


def splitmix64(x):
    return int(lib().oracle_splitmix64(x & 0xFFFFFFFFFFFFFFFF))


def tables(algo, P):
    """Preprocessing tables as numpy arrays (for diffing the GPU host builders)."""
    P = _u8(P)
    m = len(P)
    L = lib()
    if algo == "hor":
        t = np.empty(256, dtype=np.int32)
        L.oracle_pre_hor(P.ctypes.data, m, t.ctypes.data)
        return t
    if algo == "bm_gs":
        t = np.empty(m, dtype=np.int32)
        L.oracle_pre_bm_gs(P.ctypes.data, m, t.ctypes.data)
        return t
    if algo == "kmp":
        t = np.empty(m + 1, dtype=np.int32)
        L.oracle_pre_kmp(P.ctypes.data, m, t.ctypes.data)
        return t
    if algo == "so":
        t = np.empty(256, dtype=np.uint32)
        lim = L.oracle_pre_so(P.ctypes.data, m, t.ctypes.data)
        return t, int(lim)
    if algo == "bndm":
        t = np.empty(256, dtype=np.uint32)
        L.oracle_pre_bndm(P.ctypes.data, m, t.ctypes.data)
        return t
    raise ValueError(algo)


# ---------------------------------------------------------------------------
# the real reference algorithms (oracle/_ref), when built
# ---------------------------------------------------------------------------
class RefAlgo:
    """One reference algorithm built as a shared object by oracle/Makefile.

    Each reference translation unit defines the globals `run_time`, `pre_time`
    (double*) and `_timer` (TIMER*) (src/algos/include/main.h:34-37) that its
    timing macros dereference (main.h:28-31); main() normally sets them up, so
    they are pointed at scratch storage here before search() is called.
    """

    def __init__(self, name):
        path = os.path.join(REF_DIR, "lib%s.so" % name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.name = name
        self.lib = C.CDLL(path, mode=os.RTLD_LOCAL)
        self._run = C.c_double(0.0)
        self._pre = C.c_double(0.0)
        self._timer = (C.c_double * 2)()
        C.c_void_p.in_dll(self.lib, "run_time").value = C.addressof(self._run)
        C.c_void_p.in_dll(self.lib, "pre_time").value = C.addressof(self._pre)
        C.c_void_p.in_dll(self.lib, "_timer").value = C.addressof(self._timer)
        self.lib.search.restype = C.c_int
        self.lib.search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]

    def search(self, P, T, tail=None, pad=64):
        """Reference count.  The text is copied into a buffer with `pad` extra
        bytes after T[n-1] (the harness gives 10, smart.c:558; so.c:90 /
        bndm.c:101 / epsm.c:332 read further) filled with `tail`, by default a
        byte value that does not occur in P so that no straddling
        pseudo-occurrence exists (SURVEY.md §8a parity rule)."""
        P = _u8(P)
        T = _u8(T)
        m, n = len(P), len(T)
        pad = max(pad, m + 64)
        if tail is None:
            absent = np.setdiff1d(np.arange(256, dtype=np.int64), P)
            tail = int(absent[0]) if len(absent) else 0
        buf = np.full(n + pad, tail, dtype=np.uint8)
        buf[:n] = T
        pb = np.zeros(m + 8, dtype=np.uint8)  # P is NUL-terminated (smart.c:313)
        pb[:m] = P
        return int(self.lib.search(pb.ctypes.data, m, buf.ctypes.data, n))

    @property
    def run_ms(self):
        return self._run.value

    @property
    def pre_ms(self):
        return self._pre.value


def have_ref():
    return all(os.path.exists(os.path.join(REF_DIR, "lib%s.so" % a)) for a in ALGOS)
