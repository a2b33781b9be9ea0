#!/usr/bin/env python3
"""kmp_runs on the English corpus (1 GiB tiled), pattern by pattern: time, share of the roofline, K (the borderless leading states
the 0..K form covers) and the pattern — which patterns are slow and why (frequent prefixes and occurrences, not K):
    python tools/kmp_english_probe.py     (on a GPU box)"""
import sys, numpy as np
sys.path.insert(0,'.')
import smart_amd
from smart_amd import Plan, Text, corpus
from bench import PATTERN_SALT, splitmix64
unit = corpus.english_unit()
n = 1<<30
text = Text.upload_tiled(unit, n)
for m in (16, 32):
    rows=[]
    for j in range(16):
        k = splitmix64(PATTERN_SALT + 4096*j + m) % (len(unit)-m)
        P = unit[k:k+m].copy()
        tab = smart_amd.build_table("kmp_runs_compact", P).astype(np.uint8)
        thr = int(tab[-272+256:-272+260].view(np.uint32)[0])
        pl = Plan("kmp", P)
        pl.launch(text, slot=1); pl.result(1)
        ts=[]
        for r in range(3):
            pl.launch(text, slot=0, timed=True); ts.append(pl.result(0)[1])
        rows.append((min(ts), thr//4, bytes(P)))
        pl.free()
    for t,K,P in sorted(rows): print("m=%d %.4f ms  %.1f%%  K=%d  %r" % (m, t, 1.0737/t/8*100, K, P))
