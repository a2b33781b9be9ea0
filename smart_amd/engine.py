"""ctypes binding of libsmartgpu.so (include/smartgpu.h).

Mirrors SMART's plugin surface: a *text* loaded once (smart.c:553-568,95-138),
*patterns* cut from it (smart.c:148-158), and per-algorithm `search` calls that
return an occurrence count plus preprocessing / searching times in ms
(main.h:28-39).  There is no CPU fallback here: if the HIP library cannot be
loaded or no device is present, calls raise SmartGpuError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMARTGPU_LIB overrides the library path (A/B runs of two builds in one session)
LIB_PATH = os.environ.get("SMARTGPU_LIB") or os.path.join(_HERE, "csrc", "libsmartgpu.so")
# the product library plus the superseded kernels (smartgpu_tune selects them): A/B tests and measurements
AB_LIB_PATH = os.path.join(_HERE, "csrc", "libsmartgpu_ab.so")
ALGOS = ("hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita", "hash3", "hash5", "hash8", "sbndm", "kr", "bndml")
# shortest pattern each algorithm applies to (the reference returns -1 below: raita.c:37, hash3.c:31, ...)
MIN_M = {"raita": 2, "hash3": 3, "hash5": 5, "hash8": 8, "sbndm": 2}

_lib = None


class SmartGpuError(RuntimeError):
    pass


def build():
    """Compile libsmartgpu.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j", str(min(os.cpu_count() or 4, 12)), "-C", os.path.join(_HERE, "csrc")])


_loaded = {}


def use_library(path=None):
    """Make `path` (default: the product library) the library new Text / Plan objects and the module-level
    calls go to.  Objects made earlier keep the library that made them (they free through it)."""
    global _lib
    _lib = _load(path or LIB_PATH)
    return _lib


def lib():
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


def _load(path):
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise SmartGpuError("%s is missing: run `make -C smart_amd/csrc` (or __graft_entry__.build())" % path)
    # dmabuf IPC (the only mode this pool's driver supports) is an HSA flag read when the process first touches HIP:
    # set before the library — and through it ROCr — is loaded; the library's own constructor does the same for C callers
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = C.CDLL(path)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    sig = {
        "smartgpu_version": (C.c_char_p, []),
        "smartgpu_last_error": (C.c_char_p, []),
        "smartgpu_device_count": (i32, []),
        "smartgpu_algo_id": (i32, [C.c_char_p]),
        "smartgpu_algo_name": (C.c_char_p, [i32]),
        "smartgpu_device_sync": (i32, [i32]),
        "smartgpu_text_upload": (vp, [vp, u64, i32]),
        "smartgpu_text_upload_tiled": (vp, [vp, u64, u64, u64, i32]),
        "smartgpu_text_generate": (vp, [u64, i32, u64, u64, i32]),
        "smartgpu_text_free": (None, [vp]),
        "smartgpu_text_length": (u64, [vp]),
        "smartgpu_text_device": (i32, [vp]),
        "smartgpu_text_read": (i32, [vp, u64, u64, vp]),
        "smartgpu_text_alphabet": (i32, [vp, vp]),
        "smartgpu_search64": (i32, [i32, vp, u32, vp, u64, u64, C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "smartgpu_search_batch64": (i32, [i32, vp, u32, u32, vp, u64, u64, vp, vp, vp, C.POINTER(C.c_double)]),
        "smartgpu_search_batch64_each": (i32, [i32, vp, u32, u32, vp, u64, u64, vp, vp, vp, C.POINTER(C.c_double)]),
        "smartgpu_msearch_batch64": (i32, [i32, vp, u32, u32, vp, i32, vp, vp, C.POINTER(C.c_double)]),
        "smartgpu_find64": (i32, [vp, u32, vp, u64, u64, vp, u64, C.POINTER(u64)]),
        "smartgpu_last_times": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "smartgpu_plan_create": (vp, [i32, vp, u32, i32]),
        "smartgpu_plan_free": (None, [vp]),
        "smartgpu_plan_launch": (i32, [vp, vp, u64, u64, i32, i32]),
        "smartgpu_plan_result": (i32, [vp, i32, C.POINTER(u64), C.POINTER(C.c_double)]),
        "smartgpu_plan_kernel_name": (C.c_char_p, [vp]),
        "smartgpu_kernel_for": (C.c_char_p, [i32, vp, u32]),
        "smartgpu_plan_result_device_ptr": (vp, [vp]),
        "smartgpu_build_table": (i32, [i32, vp, u32, vp, u32]),
        "smartgpu_plan_reset": (i32, [vp]),
        "smartgpu_plan_set_result_buffer": (i32, [vp, vp, i32]),
        "smartgpu_stream_mark": (i32, [i32, i32]),
        "smartgpu_stream_elapsed_ms": (i32, [i32, C.POINTER(C.c_double)]),
        "smartgpu_stream_handle": (vp, [i32]),
        "smartgpu_tune": (i32, [i32, i32]),
        "smartgpu_mtext_upload": (vp, [vp, u64, i32, vp]),
        "smartgpu_mtext_generate": (vp, [u64, i32, u64, i32, vp]),
        "smartgpu_mtext_free": (None, [vp]),
        "smartgpu_mtext_length": (u64, [vp]),
        "smartgpu_mtext_ngpus": (i32, [vp]),
        "smartgpu_mtext_partition": (i32, [u64, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "smartgpu_selftest_launch_pool": (i32, [i32, i32]),
        "smartgpu_msearch64": (i32, [i32, vp, u32, vp, i32, C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "smartgpu_probe_read_ms": (i32, [vp, i32, C.POINTER(C.c_double)]),
    }
    for a in ALGOS:
        sig["smartgpu_%s_search" % a] = (i32, [vp, i32, vp, i32])
    for name, (res, args) in sig.items():
        if not hasattr(L, name) and os.path.abspath(path) != os.path.join(_HERE, "csrc", "libsmartgpu.so"):
            continue  # an older build loaded for an A/B run (SMARTGPU_LIB / use_library): it lacks the newer entry points
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _loaded[path] = L
    return L


EXPORTS = None  # filled by tests from include/smartgpu.h


def _err(what):
    return SmartGpuError("%s: %s" % (what, lib().smartgpu_last_error().decode()))


def version():
    return lib().smartgpu_version().decode()


def device_count():
    return lib().smartgpu_device_count()


def algo_id(name):
    i = lib().smartgpu_algo_id(name.encode())
    if i < 0:
        raise SmartGpuError("unknown algorithm %r" % name)
    return i


def _u8(a):
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class Text:
    """A text resident in one GPU's HBM (the shmget/getText replacement)."""

    def __init__(self, handle):
        if not handle:
            raise _err("text")
        self._h = handle
        self._L = lib()  # the library that made the handle frees it

    @classmethod
    def upload(cls, data, device=0):
        data = _u8(data)
        return cls(lib().smartgpu_text_upload(data.ctypes.data, len(data), device))

    @classmethod
    def upload_tiled(cls, unit, n, phase=0, device=0):
        unit = _u8(unit)
        return cls(lib().smartgpu_text_upload_tiled(unit.ctypes.data, len(unit), phase, n, device))

    @classmethod
    def generate(cls, seed, sigma, n, off=0, device=0):
        return cls(lib().smartgpu_text_generate(seed, sigma, off, n, device))

    def __len__(self):
        return int(lib().smartgpu_text_length(self._h))

    @property
    def device(self):
        return lib().smartgpu_text_device(self._h)

    def read(self, off, length):
        out = np.empty(length, dtype=np.uint8)
        if lib().smartgpu_text_read(self._h, off, length, out.ctypes.data) != 0:
            raise _err("text_read")
        return out

    def alphabet(self):
        """The byte values the text holds, ascending (taken on the device when the text was created)."""
        bits = np.zeros(8, dtype=np.uint32)
        if lib().smartgpu_text_alphabet(self._h, bits.ctypes.data) != 0:
            raise _err("text_alphabet")
        return [c for c in range(256) if (int(bits[c >> 5]) >> (c & 31)) & 1]

    def pattern(self, k, m):
        """P = T[k..k+m), as setOfRandomPatterns cuts it (smart.c:148-158)."""
        return self.read(k, m)

    def free(self):
        if self._h:
            self._L.smartgpu_text_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Plan:
    """One (algorithm, pattern) preprocessed and resident on a device."""

    def __init__(self, algo, P, device=0):
        self.P = _u8(P)
        self.algo = algo
        self._L = lib()
        self._h = self._L.smartgpu_plan_create(algo_id(algo), self.P.ctypes.data, len(self.P), device)
        if not self._h:
            raise _err("plan_create")

    def launch(self, text, slot=0, timed=False, off=0, n=None):
        if n is None:
            n = len(text) - off
        if lib().smartgpu_plan_launch(self._h, text._h, off, n, slot, 1 if timed else 0) != 0:
            raise _err("plan_launch")

    def result(self, slot=0):
        c = C.c_uint64(0)
        ms = C.c_double(0.0)
        if lib().smartgpu_plan_result(self._h, slot, C.byref(c), C.byref(ms)) != 0:
            raise _err("plan_result")
        return int(c.value), float(ms.value)

    def reset(self):
        if lib().smartgpu_plan_reset(self._h) != 0:
            raise _err("plan_reset")

    def set_result_buffer(self, device_ptr, nslots=1):
        if lib().smartgpu_plan_set_result_buffer(self._h, device_ptr, nslots) != 0:
            raise _err("plan_set_result_buffer")

    @property
    def kernel_name(self):
        return lib().smartgpu_plan_kernel_name(self._h).decode()

    @property
    def result_device_ptr(self):
        return lib().smartgpu_plan_result_device_ptr(self._h)

    def free(self):
        if self._h:
            self._L.smartgpu_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MultiText:
    """A text sharded over several GPUs of this process (smartgpu_mtext_*)."""

    def __init__(self, handle):
        if not handle:
            raise _err("mtext")
        self._h = handle
        self._L = lib()

    @staticmethod
    def _devs(devices):
        if devices is None:
            return None, None
        arr = (C.c_int * len(devices))(*devices)
        return arr, C.cast(arr, C.c_void_p)

    @classmethod
    def upload(cls, data, ngpus, devices=None):
        data = _u8(data)
        keep, ptr = cls._devs(devices)
        return cls(lib().smartgpu_mtext_upload(data.ctypes.data, len(data), ngpus, ptr))

    @classmethod
    def generate(cls, seed, sigma, n, ngpus, devices=None):
        keep, ptr = cls._devs(devices)
        return cls(lib().smartgpu_mtext_generate(seed, sigma, n, ngpus, ptr))

    def __len__(self):
        return int(lib().smartgpu_mtext_length(self._h))

    def search(self, algo, P, reduce="rccl"):
        P = _u8(P)
        c = C.c_uint64(0)
        pre = C.c_double(0.0)
        run = C.c_double(0.0)
        rc = lib().smartgpu_msearch64(algo_id(algo), P.ctypes.data, len(P), self._h, 0 if reduce == "rccl" else 1,
                                      C.byref(c), C.byref(pre), C.byref(run))
        if rc != 0:
            raise _err("msearch64(%s) rc=%d" % (algo, rc))
        return int(c.value), float(pre.value), float(run.value)

    def search_batch(self, algo, patterns, reduce="rccl"):
        """(counts, pre_ms, batch_ms): every shard searched for all K patterns, ONE reduction of the K counts."""
        pats, ptrs, m = _pattern_set(patterns)
        K = len(pats)
        counts = np.zeros(K, dtype=np.uint64)
        pre = np.zeros(K, dtype=np.float64)
        batch = C.c_double(0.0)
        rc = lib().smartgpu_msearch_batch64(algo_id(algo), C.cast(ptrs, C.c_void_p), m, K, self._h, 0 if reduce == "rccl" else 1,
                                            counts.ctypes.data, pre.ctypes.data, C.byref(batch))
        if rc != 0:
            raise _err("msearch_batch64(%s) rc=%d" % (algo, rc))
        return counts, pre, float(batch.value)

    def free(self):
        if self._h:
            self._L.smartgpu_mtext_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def mtext_partition(n, ngpus, g):
    """(begin, own, held) of shard g of a text of n bytes over ngpus devices — smartgpu_mtext_partition, the arithmetic
    smartgpu_mtext_upload / _generate shard with; no device needed."""
    b, o, h = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    if lib().smartgpu_mtext_partition(n, ngpus, g, C.byref(b), C.byref(o), C.byref(h)) != 0:
        raise _err("mtext_partition")
    return int(b.value), int(o.value), int(h.value)


def search(algo, P, text, off=0, n=None):
    """(count, pre_ms, run_ms) of `algo` for P in text[off..off+n)."""
    P = _u8(P)
    if n is None:
        n = len(text) - off
    c = C.c_uint64(0)
    pre = C.c_double(0.0)
    run = C.c_double(0.0)
    rc = lib().smartgpu_search64(algo_id(algo), P.ctypes.data, len(P), text._h, off, n,
                                 C.byref(c), C.byref(pre), C.byref(run))
    if rc != 0:
        raise _err("search64(%s) rc=%d" % (algo, rc))
    return int(c.value), float(pre.value), float(run.value)


def _pattern_set(patterns):
    """K patterns of one length as the const uint8_t* const* the batch calls take (+ what must stay alive)."""
    pats = [_u8(p) for p in patterns]
    m = len(pats[0])
    if any(len(p) != m for p in pats):
        raise SmartGpuError("a pattern set holds patterns of ONE length (smart.c:312 loops per length)")
    ptrs = (C.c_void_p * len(pats))(*[p.ctypes.data for p in pats])
    return pats, ptrs, m


def search_batch(algo, patterns, text, off=0, n=None, per_pattern_times=True, each=False):
    """(counts, pre_ms, run_ms, batch_ms) of `algo` for a whole pattern set over text[off..off+n): the
    harness loop of smart.c:312-345 as one call (smartgpu_search_batch64).  run_ms is None without
    per_pattern_times.  each: every pattern its own launch and event pair (smartgpu_search_batch64_each)."""
    pats, ptrs, m = _pattern_set(patterns)
    K = len(pats)
    if n is None:
        n = len(text) - off
    counts = np.zeros(K, dtype=np.uint64)
    pre = np.zeros(K, dtype=np.float64)
    run = np.zeros(K, dtype=np.float64) if per_pattern_times else None
    batch = C.c_double(0.0)
    fn = lib().smartgpu_search_batch64_each if each else lib().smartgpu_search_batch64
    rc = fn(algo_id(algo), C.cast(ptrs, C.c_void_p), m, K, text._h, off, n, counts.ctypes.data,
            pre.ctypes.data, run.ctypes.data if run is not None else None, C.byref(batch))
    if rc != 0:
        raise _err("search_batch64(%s) rc=%d" % (algo, rc))
    return counts, pre, run, float(batch.value)


def find(P, text, off=0, n=None, cap=1 << 20):
    """(positions, count): the ascending start offsets (relative to text byte 0) of P in
    text[off..off+n), and their number.  When there are more than `cap`, positions is None and only
    the count is returned (smartgpu_find64 reports SMARTGPU_ERR_NOMEM; retry with cap >= count)."""
    P = _u8(P)
    if n is None:
        n = len(text) - off
    out = np.empty(max(cap, 1), dtype=np.uint64)
    c = C.c_uint64(0)
    rc = lib().smartgpu_find64(P.ctypes.data, len(P), text._h, off, n, out.ctypes.data, cap, C.byref(c))
    if rc == -5 and c.value > cap:
        return None, int(c.value)
    if rc != 0:
        raise _err("find64 rc=%d" % rc)
    return out[:c.value].copy(), int(c.value)


def search_host(algo, P, T):
    """SMART's own `int search(P, m, T, n)` shape on host buffers."""
    P = _u8(P)
    T = _u8(T)
    return getattr(lib(), "smartgpu_%s_search" % algo)(P.ctypes.data, len(P), T.ctypes.data, len(T))


def device_sync(device=0):
    if lib().smartgpu_device_sync(device) != 0:
        raise _err("device_sync")


def stream_mark(device, which):
    if lib().smartgpu_stream_mark(device, which) != 0:
        raise _err("stream_mark")


def stream_elapsed_ms(device):
    ms = C.c_double(0.0)
    if lib().smartgpu_stream_elapsed_ms(device, C.byref(ms)) != 0:
        raise _err("stream_elapsed_ms")
    return float(ms.value)


def probe_read_gbs(text, reps=20):
    """Practical streaming-read rate (GB/s) of the device on this text."""
    ms = C.c_double(0.0)
    if lib().smartgpu_probe_read_ms(text._h, reps, C.byref(ms)) != 0:
        raise _err("probe_read")
    return len(text) / (ms.value * 1e-3) / 1e9


def tune(key, value):
    if lib().smartgpu_tune(key, value) != 0:
        raise _err("tune")


def kernel_for(algo, P):
    """Kernel a plan of (algo, P) would launch under the current tune settings; no device needed."""
    P = _u8(P)
    name = lib().smartgpu_kernel_for(algo_id(algo), P.ctypes.data, len(P))
    if name is None:
        raise _err("kernel_for")
    return name.decode()


def build_table(which, P):
    P = _u8(P)
    names = {"bad_char": 0, "good_suffix": 1, "kmp_next": 2, "shift_or": 3, "bndm": 4, "kmp_dfa": 5,
             "kmp_dfa_compressed": 6, "shift_and": 7, "quick_search": 8, "kmp_runs": 9, "four_codes": 10, "kmp_runs_compact": 11, "hash3": 13, "hash5": 15, "hash8": 18}
    out = np.empty(max(257, len(P) + 1, (len(P) + 1) * 256 + 257 if which.startswith("kmp_dfa") else 256 * 256 + 272 if which.startswith("kmp_runs") else 0), dtype=np.int32)
    k = lib().smartgpu_build_table(names[which], P.ctypes.data, len(P), out.ctypes.data, len(out))
    if k < 0:
        raise _err("build_table")
    return out[:k].copy()
