"""smart_amd — MI355X-native exact string matching behind SMART's `search()` plugin surface.

The product is the C-ABI shared library `smart_amd/csrc/libsmartgpu.so`
(include/smartgpu.h); this package is the thin ctypes binding plus a Python
mirror of SMART's harness vocabulary (texts, patterns, algorithms).
"""
from .engine import (ALGOS, MIN_M, MultiText, Plan, SmartGpuError, Text, algo_id, build_table, device_count, kernel_for,  # noqa: F401
                     find, lib, search, search_batch, search_host, version)
