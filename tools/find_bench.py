#!/usr/bin/env python3
"""Wall time of smartgpu_find64 (plan + kernel + D2H of the positions + host sort) on the
BASELINE-style corpora: python tools/find_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smart_amd  # noqa: E402
from smart_amd import Text  # noqa: E402
from bench import SEED  # noqa: E402

n = 1 << 30
for sigma, ms in ((128, (2, 4, 8, 32, 256)), (4, (4, 8, 16, 64)), (2, (16, 32, 64))):
    text = Text.generate(SEED, sigma, n)
    for m in ms:
        P = text.pattern(123457, m)
        cnt = smart_amd.search("epsm", P, text)[0]
        cap = max(cnt, 1)
        smart_amd.find(P, text, cap=cap)  # warm-up
        t0 = time.perf_counter()
        pos, c = smart_amd.find(P, text, cap=cap)
        dt = time.perf_counter() - t0
        assert c == cnt and len(pos) == cnt and (pos[1:] > pos[:-1]).all()
        print("find  sigma=%-3d m=%-4d %10d positions  %8.3f ms  %7.1f GB/s of text" % (sigma, m, cnt, dt * 1e3, n / dt / 1e9), flush=True)
    text.free()
