#!/usr/bin/env python3
"""bench.py — headline benchmark of the exact-matching hot path on MI355X.

Metric (BASELINE.json): GB/s of text scanned, bit-exact occurrence count,
Horspool m=32 on 1 GiB of rand128 per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--algo hor --plen 32 --sigma 128 --gib 1]

One "step" = one search: one pattern (cut from the text at a seeded offset, as
setOfRandomPatterns does, src/smart.c:148-158) scanned over the whole resident
text.  The text is generated on the device (counter-based rand-sigma corpus,
SURVEY.md §8d) and is resident in HBM before the timed region; every step's
preprocessing (table build + upload = SMART's pre_time) is done before it as
well, so the timed region is SMART's run_time: kernels + the reduction of the
counts.  With N>1 ranks (one per GPU, torchrun) the text is N GiB sharded by
byte offset with an (m-1)-byte overlap; the K counts are summed with ONE RCCL
all-reduce inside the timed region ("scaling": "weak").

After the timed region every count is checked: against an independent kernel
(EPSM packed matcher) for all K patterns and against the CPU oracle / the real
reference build for the cpu_baseline sample.  A mismatch aborts the run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5EED0001        # corpus seed
PATTERN_SALT = 0x0A77E2  # k_j = splitmix64(PATTERN_SALT + 4096*j + m) mod (n-m)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def splitmix64(x):
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def load_traffic(kernel, workload_key):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/),
    already corrected as MI355X_MICROARCH.md §HBM prescribes; None if absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        return table.get(kernel, {}).get(workload_key)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--algo", default="hor")
    ap.add_argument("--plen", dest="m", type=int, default=32, help="pattern length m")
    ap.add_argument("--sigma", type=int, default=128)
    ap.add_argument("--gib", type=float, default=1.0, help="text GiB per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo); the numbers mean nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))

    import torch
    import smart_amd
    from smart_amd import Plan, Text, engine

    if args.share_gpu:
        local_rank = 0
    if smart_amd.device_count() <= local_rank:
        raise SystemExit("no GPU for local rank %d: %s" % (local_rank, engine.lib().smartgpu_last_error().decode()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    on_host = dist is not None and args.backend != "nccl"  # gloo rehearsal reduces host copies

    def all_reduce_counts(t, op=None):
        if dist is None:
            return t
        if on_host:
            h = t.cpu()
            dist.all_reduce(h) if op is None else dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t) if op is None else dist.all_reduce(t, op=op)
        return t

    K, W, m, algo = args.steps, args.warmup, args.m, args.algo
    shard = int(args.gib * (1 << 30))
    total_n = shard * world
    # rank r owns start positions [r*shard, (r+1)*shard) (the last rank stops at
    # total_n - m); it holds m-1 extra bytes so those windows are complete
    from smart_amd.sharding import weak_shard
    global_off, local_len = weak_shard(shard, m, rank, world)
    text = Text.generate(SEED, args.sigma, local_len, off=global_off, device=local_rank)

    # patterns: cut from the GLOBAL text at seeded offsets (any shard)
    def pattern(j):
        k = splitmix64(PATTERN_SALT + 4096 * j + m) % (total_n - m)
        t = Text.generate(SEED, args.sigma, m, off=k, device=local_rank)
        p = t.read(0, m)
        t.free()
        return k, p

    pats = [pattern(j) for j in range(W + K)]
    counts = torch.zeros(W + K, dtype=torch.int64, device="cuda")
    plans = []
    t_pre = time.perf_counter()
    for j, (_, p) in enumerate(pats):
        pl = Plan(algo, p, device=local_rank)   # preprocessing: tables built + placed in HBM
        pl.set_result_buffer(counts.data_ptr() + 8 * j, 1)
        plans.append(pl)
    pre_ms = (time.perf_counter() - t_pre) * 1e3 / (W + K)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(lo, hi, mark=False):
        if mark:
            engine.stream_mark(local_rank, 0)
        for j in range(lo, hi):
            plans[j].launch(text, slot=0)
        if mark:
            engine.stream_mark(local_rank, 1)
        engine.device_sync(local_rank)          # counts are in HBM
        all_reduce_counts(counts[lo:hi])       # ONE RCCL sum of the K counts over xGMI
        return counts[lo:hi].cpu()             # ... and on the host (SMART's run_time ends here)

    barrier()
    run(0, W)
    barrier()
    t0 = time.perf_counter()
    got = run(W, W + K, mark=True)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = engine.stream_elapsed_ms(local_rank) / K   # HIP events on the launch stream

    if dist is not None:
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        all_reduce_counts(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tt[0]), float(tt[1])

    # ---- verification (outside the timed region) -------------------------------
    got = got.numpy().astype(np.uint64)
    check = torch.zeros(K, dtype=torch.int64, device="cuda")
    other = "epsm" if algo != "epsm" else "hor"
    oplans = []
    for j in range(K):
        pl = Plan(other, pats[W + j][1], device=local_rank)
        pl.set_result_buffer(check.data_ptr() + 8 * j, 1)
        pl.launch(text, slot=0)
        oplans.append(pl)
    engine.device_sync(local_rank)
    all_reduce_counts(check)
    want = check.cpu().numpy().astype(np.uint64)
    if not np.array_equal(got, want):
        raise SystemExit("COUNT MISMATCH %s vs %s: %s != %s" % (algo, other, got.tolist(), want.tolist()))
    if int(got.min()) < 1:
        raise SystemExit("a planted pattern was not found: %s" % got.tolist())

    # ---- CPU baseline (rank 0, N=1 only): SMART's own algorithm on host cores ---
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle
        pyoracle.build(ref=False)
        T = text.read(0, local_len)
        assert np.array_equal(T[: 1 << 16], pyoracle.gen_text(SEED, args.sigma, 0, 1 << 16))
        ref = None
        if algo in pyoracle.ALGOS and pyoracle.have_ref() and local_len < (1 << 31):
            ref = pyoracle.RefAlgo(algo)
        nsample = min(K, 6)
        t_cpu = 0.0
        for j in range(nsample):
            P = pats[W + j][1]
            if ref is not None:
                buf = np.zeros(local_len + m + 64, dtype=np.uint8)  # reference reads past T[n-1] (so.c:90)
                buf[:local_len] = T
                t1 = time.perf_counter()
                c = ref.lib.search(P.ctypes.data, m, buf.ctypes.data, local_len)
                t_cpu += time.perf_counter() - t1
            else:
                t1 = time.perf_counter()
                c = pyoracle.search(algo, P, T)
                t_cpu += time.perf_counter() - t1
            if int(c) != int(got[j]):
                raise SystemExit("COUNT MISMATCH vs CPU %s: pattern %d gpu %d cpu %d" % (algo, j, got[j], c))
        cores = os.cpu_count() or 1
        t2 = time.perf_counter()
        cmt = pyoracle.search(algo, pats[W][1], T, threads=cores)
        t_mt = time.perf_counter() - t2
        assert int(cmt) == int(got[0])
        cpu = {
            "value": round(nsample * local_len / t_cpu / 1e9, 3), "unit": "GB/s", "cores": 1,
            "kind": "reference" if ref is not None else "port",
            "sample": "%d of the %d timed patterns over the full %.2f GiB text, single thread (SMART is single-threaded)"
                      % (nsample, K, local_len / 2**30),
            "all_cores": {"value": round(local_len / t_mt / 1e9, 3), "cores": cores, "kind": "port",
                          "sample": "1 pattern, text split by core with (m-1) overlap"},
        }

    # per-pattern spread of the kernel time, as SMART reports mean/best/worst/std over a pattern set
    # (smart.c:347-351): a second, untimed pass over the same K plans with one event pair per launch
    spread = None
    if rank == 0:
        for j in range(W, W + K):
            plans[j].launch(text, slot=0, timed=True)
        per = np.array([plans[j].result(0)[1] for j in range(W, W + K)])
        spread = {"mean": round(float(per.mean()), 4), "best": round(float(per.min()), 4),
                  "worst": round(float(per.max()), 4), "std": round(float(per.std()), 4),
                  "note": "ms per pattern, separate pass after the timed region, HIP events per launch"}

    read_probe = engine.probe_read_gbs(text) if rank == 0 else None
    if rank == 0:
        bytes_per_launch = local_len                       # algorithmic bytes: every text byte once
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        workload = "%s m=%d, %.2f GiB rand%d per GPU" % (algo.upper(), m, args.gib, args.sigma)
        traffic = load_traffic(plans[0].kernel_name, "%s_m%d_sigma%d_gib%g" % (algo, m, args.sigma, args.gib))
        out = {
            "metric": "GB/s text scanned per GPU (bit-exact occ count), m=32 on 1 GiB rand128",
            "value": round(total_n * K / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed * 1e3 / K, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "algorithm": algo, "m": m, "sigma": args.sigma,
                       "text_bytes_per_gpu": local_len, "patterns": K,
                       "corpus": "counter-based splitmix64 rand-sigma, seed 0x5EED0001, generated on device",
                       "sharding": "byte offset, (m-1) overlap, one RCCL all-reduce of the K counts" if world > 1 else "single GPU",
                       "pre_ms_per_pattern": round(pre_ms, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": plans[0].kernel_name, "kernel_ms": round(kernel_ms, 4),
                         "bytes_per_launch": bytes_per_launch, "kernel_ms_per_pattern": spread,
                         "measured_stream_read_GBps": round(read_probe, 1),
                         "frac_of_measured_stream_read": round(achieved / read_probe, 4)},
            "cpu_baseline": cpu,
            "counts_verified": "all %d counts equal the %s kernel%s" % (K, other, " and the CPU sample" if cpu else ""),
        }
        print(json.dumps(out))

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
