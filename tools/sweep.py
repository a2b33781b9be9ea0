#!/usr/bin/env python3
"""Kernel-time sweep over (algorithm, m, sigma) on one GPU — no torch.

    python tools/sweep.py [--gib 1] [--sigma 128] [--ms 4,8,32,256] [--algos hor,bm,...] [--reps 5]

Prints one line per (algo, m): median device time of the scan kernel (HIP
events on the launch stream), GB/s of text scanned and the fraction of the
8 TB/s HBM peak; checks that all algorithms return the same count per pattern.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smart_amd  # noqa: E402
from smart_amd import Plan, Text  # noqa: E402

sys.path.insert(0, ROOT)
from bench import PATTERN_SALT, SEED, splitmix64  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--sigma", type=int, default=128)
    ap.add_argument("--ms", default="4,8,32,256")
    ap.add_argument("--algos", default=",".join(smart_amd.ALGOS[:6]), help="default: the six hot-path algorithms")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--tune", default="", help="comma list key=value passed to smartgpu_tune")
    ap.add_argument("--corpus", default="rand", help="rand (counter-based rand<sigma>) or english "
                    "(bible.txt||world192.txt, 6,520,792 bytes, tiled to --gib: BASELINE config 4)")
    ap.add_argument("--own", action="store_true", help="smartgpu_tune(0,1): every algorithm on its own kernel, no rerouting")
    args = ap.parse_args()
    for kv in filter(None, args.tune.split(",")):
        k, v = kv.split("=")
        smart_amd.engine.tune(int(k), int(v))
    if args.own:
        smart_amd.engine.tune(0, 1)
    n = int(args.gib * (1 << 30))
    unit = None
    if args.corpus == "english":
        from smart_amd import corpus
        unit = corpus.english_unit()  # the whole corpus as getText loads it, not an excerpt (round 1 tiled 262,139 bytes)
        text = Text.upload_tiled(unit, n)
    else:
        text = Text.generate(SEED, args.sigma, n)
    print("streaming-read probe: %.1f GB/s" % smart_amd.engine.probe_read_gbs(text), flush=True)
    rows = []
    for m in [int(x) for x in args.ms.split(",")]:
        pats = []
        for j in range(args.reps):
            k = splitmix64(PATTERN_SALT + 4096 * j + m) % ((len(unit) if unit is not None else n) - m)  # English: from the first copy
            pats.append(text.pattern(k, m))
        ref_counts = None
        for algo in args.algos.split(","):
            if m < smart_amd.MIN_M.get(algo, 1):  # the algorithm does not apply (raita.c:37, hash3.c:31, ...)
                continue
            plans = [Plan(algo, p) for p in pats]
            kernels = sorted({pl.kernel_name for pl in plans})  # the plan's choice (api.cpp build_blob), per pattern
            plans[0].launch(text, slot=1)  # warm-up
            plans[0].result(1)
            for pl in plans:
                pl.launch(text, slot=0, timed=True)
            res = [pl.result(0) for pl in plans]
            counts = [c for c, _ in res]
            times = sorted(t for _, t in res)
            if ref_counts is None:
                ref_counts = counts
            ok = counts == ref_counts
            med = times[len(times) // 2]
            gbs = n / (med * 1e-3) / 1e9
            row = {"algo": algo, "m": m, "sigma": args.sigma, "n": n, "kernel_ms_median": round(med, 4),
                   "kernel_ms_min": round(times[0], 4), "GBps": round(gbs, 1), "frac_hbm_peak": round(gbs / 8000.0, 4),
                   "counts": counts, "counts_agree": ok, "kernels": kernels}
            rows.append(row)
            print("%-5s m=%-5d sigma=%-3d  %8.4f ms (min %8.4f)  %8.1f GB/s  %5.1f%% of 8 TB/s  counts %s %s [%s]"
                  % (algo, m, args.sigma, med, times[0], gbs, gbs / 80.0, counts[:3], "" if ok else "MISMATCH", "+".join(kernels)),
                  flush=True)
            for pl in plans:
                pl.free()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)
    if not all(r["counts_agree"] for r in rows):
        raise SystemExit("count mismatch between algorithms")


if __name__ == "__main__":
    main()
