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
counts.

N > 1: one process per GPU.  Started plainly (`python bench.py --gpus N`, no
WORLD_SIZE in the environment) this process launches the N ranks itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ...` BEFORE anything touches a GPU, relays their output and exits with
their code; started under torchrun (the driver's way) it is one of the ranks.
The text is N x --gib GiB sharded by byte offset with an (m-1)-byte overlap; the
K counts are summed with ONE RCCL all-reduce inside the timed region
("scaling": "weak").  `--gib 4` gives every GPU the 4 GiB shard of BASELINE
configs 4-5; `--scaling strong` splits ONE text of --gib GiB over the ranks.

After the timed region every count is checked: against a kernel of a different
family (the KMP automaton when the timed plans run on the packed matcher, the
packed matcher otherwise) for all K patterns and against the real reference
build / the CPU oracle for the cpu_baseline sample.  A mismatch aborts the run.

At N = 1 a per-cell sweep follows: every cell of BASELINE config 2
(HOR/BM/KMP/SO/BNDM/EPSM x m in {4,8,32,256} on the 1 GiB rand128 text), of
config 3 (SO, BNDM x sigma in {2,4} x m in {2..64} on 1 GiB) and of configs 4 and
5 at the 4 GiB BASELINE.json states per GPU, each timed with HIP events (the median of 4 rounds of 3 patterns), with the kernel that
ran, its fraction of the 8 TB/s HBM peak and a count check; cells whose plans
were rerouted to another kernel are measured again on the algorithm's own kernel
(smartgpu_tune(0,1)).  The CELLS go to a file (--sweep-out, default
bench_sweep.json next to this script) — as the reference prints one summary line
per algorithm and keeps the details in files (src/smart.c:347-378) — and the one
JSON line on stdout carries only their summary ("min_frac", "own_kernel_min"):
the line stays under 4 KB (compact_line), the driver keeps 8 KB of stdout.
"""
import argparse
import collections
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x5EED0001        # corpus seed
PATTERN_SALT = 0x0A77E2  # k_j = splitmix64(PATTERN_SALT + 4096*j + m) mod (n-m)
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BIG_GIB = 4.0            # run_sweep: the text of BASELINE configs 4 and 5 per GPU (4 GiB English; one of eight 4 GiB shards)
OWN_KERNEL = {"hor": "hor_scan", "bm": "bm_scan", "kmp": "kmp_runs", "so": "so_runs", "bndm": "bndm_scan",
              "epsm": "packed_scan"}


def splitmix64(x):
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
    return x ^ (x >> 31)


def kernel_sha256(kernel):
    """sha256 of the sources that decide `kernel`'s code: its family's translation unit + the common headers
    (smart_amd/sources.py) — not of every kernel's source."""
    from smart_amd import sources
    return sources.kernel_sha256(kernel)


def load_traffic(kernel, workload_key, path=None):
    """(bytes, source) — HBM bytes per launch from the committed rocprofv3 --pmc pass (profiles/),
    corrected as MI355X_MICROARCH.md §HBM prescribes.  The figure is only valid for the kernel
    source it was profiled on: profiles/pmc_traffic.json records, per kernel, the sha256 of that
    kernel FAMILY's sources (one translation unit per family), and a different source gives (None, why)."""
    path = path or os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json is missing"
    src = table.get("_source", {})
    entry = table.get(kernel)
    if entry is None:
        return None, "no PMC pass for %s" % kernel
    try:
        have = kernel_sha256(kernel)
    except KeyError:
        return None, "no source set known for kernel %s" % kernel
    if entry.get("_sha256") != have:
        return None, ("not measured for this kernel source: %s's sources sha256 %s..., the PMC pass in %s was taken on %s..."
                      % (kernel, have[:12], src.get("summary", "profiles/"), str(entry.get("_sha256"))[:12]))
    v = entry.get(workload_key)
    if v is None:
        return None, "no PMC pass for %s / %s" % (kernel, workload_key)
    return v, "%s @ %s (%s sources sha256 %s...)" % (src.get("summary", "profiles/"), src.get("commit", "?"), kernel, have[:12])


def launch_ranks(n):
    """Plain `python bench.py --gpus N`: start the N ranks as children through torch.distributed.run.
    Nothing in this process has touched a GPU (no HIP call, no torch import)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    return subprocess.call(cmd, env=env)


def main():
    # dmabuf IPC only on this pool: RCCL between ranks needs it, and the driver's torchrun starts the ranks
    # without going through launch_ranks() — so every entry sets it, before torch or HIP is loaded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--algo", default="hor")
    ap.add_argument("--plen", dest="m", type=int, default=32, help="pattern length m")
    ap.add_argument("--sigma", type=int, default=128)
    ap.add_argument("--gib", type=float, default=1.0, help="text GiB per GPU (--scaling strong: of the whole text)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (the contract's default): --gib per GPU; strong: one text of --gib GiB split over the ranks")
    ap.add_argument("--corpus", default="rand", help="rand (counter-based rand<sigma>) or english "
                    "(tests/golden/english_bible_world192.txt.xz tiled to --gib per GPU, BASELINE config 4)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-sweep", action="store_true", help="skip the per-cell sweep of configs 2 to 5")
    ap.add_argument("--sweep-out", default=os.path.join(ROOT, "bench_sweep.json"),
                    help="file the sweep's cells are written to (the stdout line carries their summary only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo); the numbers mean nothing")
    ap.add_argument("--check-launch", action="store_true",
                    help="ranks only rendezvous (gloo), all-reduce their rank ids and print what they see; no GPU needed")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    if args.check_launch:
        import torch
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        if rank == 0:
            print(json.dumps({"check_launch": True, "n_gpus": world, "rank_id_sum": int(t[0]),
                              "launched_by": "torch.distributed.run", "master_addr": os.environ.get("MASTER_ADDR")}))
        if world > 1:
            dist.destroy_process_group()
        return

    import numpy as np
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
    shard = int(args.gib * (1 << 30)) // (world if args.scaling == "strong" else 1)
    total_n = shard * world
    # rank r owns start positions [r*shard, (r+1)*shard) (the last rank stops at
    # total_n - m); it holds m-1 extra bytes so those windows are complete
    from smart_amd.sharding import weak_shard
    global_off, local_len = weak_shard(shard, m, rank, world)
    english = None
    if args.corpus == "english":
        from smart_amd import corpus
        english = corpus.english_unit()
        text = Text.upload_tiled(english, local_len, phase=global_off % len(english), device=local_rank)
    else:
        text = Text.generate(SEED, args.sigma, local_len, off=global_off, device=local_rank)

    # patterns: cut from the GLOBAL text at seeded offsets (any shard); English: from the first copy
    def pattern(j):
        if english is not None:
            k = splitmix64(PATTERN_SALT + 4096 * j + m) % (len(english) - m)
            return k, english[k:k + m].copy()
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

    # The plans above were built with the GPU idle (host table construction, small copies): bring the chip back
    # to its working clocks before the W warm-up steps, or a short timed region measures the ramp, not the scan
    # (20 steps = 3 ms; observed: 0.173 ms per step right after an idle phase, 0.156 ms in steady state).
    PREWARM_PASSES = 64   # streaming reads of the resident text, ~10 ms; not steps, nothing of the timed work
    engine.probe_read_gbs(text, reps=PREWARM_PASSES)
    barrier()
    run(0, W)
    barrier()
    t0 = time.perf_counter()
    got = run(W, W + K, mark=True)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = engine.stream_elapsed_ms(local_rank) / K   # HIP events on the launch stream

    group = None
    if dist is not None:
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        all_reduce_counts(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tt[0]), float(tt[1])
        # what the LIVE process group consists of (not what the environment promised): every rank's device as the
        # runtime names it — the record proves N distinct GPUs behind the one all-reduce, or says that it was a rehearsal
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": dist.get_rank(), "local_rank": local_rank, "host": socket.gethostname(), "device": props.name,
                "uuid": str(getattr(props, "uuid", "")), "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0),
                                                                                    getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))}
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        ids = [(e["host"], e["uuid"] or e["pci"]) for e in everyone]
        group = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "distinct_devices": len(set(ids)),
                 "devices": [[e["rank"], e["pci"], e["uuid"][-12:]] for e in everyone]}

    # ---- verification (outside the timed region) -------------------------------
    got = got.numpy().astype(np.uint64)
    kernel_hist = collections.Counter(plans[j].kernel_name for j in range(W, W + K))
    main_kernel = kernel_hist.most_common(1)[0][0]
    check = torch.zeros(K, dtype=torch.int64, device="cuda")
    others = []
    oplans = []
    for j in range(K):
        # a kernel of another family than the one that produced the count
        # (the packed matcher, the KMP automaton, Shift-Or: the first whose plan for THIS pattern leads elsewhere —
        # on a binary text EPSM's own plan counts on so_runs too)
        pl = None
        for other in ("epsm", "kmp", "so"):
            if other == algo:
                continue
            cand = Plan(other, pats[W + j][1], device=local_rank)
            if cand.kernel_name != plans[W + j].kernel_name:
                pl = cand
                break
            cand.free()
        assert pl is not None, (plans[W + j].kernel_name, algo)
        others.append(other)
        pl.set_result_buffer(check.data_ptr() + 8 * j, 1)
        pl.launch(text, slot=0)
        oplans.append(pl)
    engine.device_sync(local_rank)
    all_reduce_counts(check)
    want = check.cpu().numpy().astype(np.uint64)
    if not np.array_equal(got, want):
        raise SystemExit("COUNT MISMATCH %s vs %s: %s != %s" % (algo, sorted(set(others)), got.tolist(), want.tolist()))
    if int(got.min()) < 1:
        raise SystemExit("a planted pattern was not found: %s" % got.tolist())
    for pl in oplans:
        pl.free()

    # ---- CPU baseline (rank 0, N=1 only): SMART's own algorithm on host cores ---
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle
        pyoracle.build(ref=False)
        T = text.read(0, local_len)
        if english is None:
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
        del T

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
    for pl in plans:
        pl.free()

    # ---- per-cell sweep of BASELINE configs 2 and 3 (N = 1) ---------------------
    sweep = None
    if rank == 0 and world == 1 and not args.no_sweep and english is None:
        sweep = run_sweep(text if (args.sigma == 128 and local_len == 1 << 30) else None, local_rank)
    text.free()

    if rank == 0:
        bytes_per_launch = local_len                       # algorithmic bytes: every text byte once
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        if english is not None:
            workload = "%s m=%d, English (bible.txt||world192.txt, %d B) tiled to %.2f GiB per GPU" % (algo.upper(), m, len(english), shard / (1 << 30))
            key = "%s_m%d_english_gib%g" % (algo, m, args.gib)
        else:
            workload = "%s m=%d, %.2f GiB rand%d per GPU" % (algo.upper(), m, shard / (1 << 30), args.sigma)
            key = "%s_m%d_sigma%d_gib%g" % (algo, m, args.sigma, args.gib)
        traffic, traffic_source = load_traffic(main_kernel, key)
        out = {
            "metric": "GB/s text scanned per GPU (bit-exact occ count), m=32 on 1 GiB rand128",
            # the contract's value is the WHOLE JOB (all N GPUs); the metric is quoted per GPU: value_per_gpu says that one
            "value": round(total_n * K / elapsed / 1e9, 2),
            "value_per_gpu": round(total_n * K / elapsed / 1e9 / world, 2),
            "aggregate": round(total_n * K / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed * 1e3 / K, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if english is None else "reference corpus (englishTexts), tiled",
            "config": {"workload": workload, "algorithm": algo, "m": m, "sigma": args.sigma if english is None else None,
                       "text_bytes_per_gpu": local_len, "patterns": K,
                       "corpus": ("counter-based splitmix64 rand-sigma, seed 0x5EED0001, generated on device" if english is None
                                  else "bible.txt||world192.txt as getText loads it (smart.c:95-138), tiled on device"),
                       "sharding": ("byte offset, (m-1) overlap, one all-reduce of the K counts over %d ranks (%s)"
                                    % (world, "RCCL" if args.backend == "nccl" else args.backend + ", rehearsal")) if world > 1 else "single GPU",
                       "ranks": group["world_size"] if group else 1,  # of the live process group
                       "pre_ms_per_pattern": round(pre_ms, 4),
                       "prewarm": "%d streaming-read passes over the text before the warm-up steps (clocks), untimed" % PREWARM_PASSES},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": main_kernel, "kernels_of_the_timed_plans": dict(kernel_hist),
                         "kernel_ms": round(kernel_ms, 4),
                         "bytes_per_launch": bytes_per_launch, "kernel_ms_per_pattern": spread,
                         "measured_stream_read_GBps": round(read_probe, 1),
                         "frac_of_measured_stream_read": round(achieved / read_probe, 4)},
            "cpu_baseline": cpu,
            "rccl": group,  # N > 1: the live group — backend, world size, every rank's device (PCI id, uuid tail)
            "counts_verified": "all %d counts equal a kernel of another family (%s)%s"
                               % (K, "/".join(sorted(set(others))), " and the CPU sample" if cpu else ""),
        }
        if sweep is not None:
            out["roofline_secondary"] = sweep["roofline_secondary"]
            out["min_frac"] = sweep["min_frac"]
            out["own_kernel_min"] = sweep["own_kernel_min"]
            out["worst_cells"] = sweep["worst_cells"]
            out["sweep_cells"] = len(sweep["cells"])
            try:
                body = json.dumps({"headline": out, "note": sweep["note"], "cells": sweep["cells"]})
                with open(args.sweep_out, "w") as f:
                    f.write(body)
                out["sweep_file"] = os.path.relpath(args.sweep_out, ROOT)
                out["sweep_sha256"] = hashlib.sha256(body.encode()).hexdigest()  # a pulled copy of the file can be matched to this line
            except OSError as e:   # a read-only checkout: the summary is in the line all the same
                out["sweep_file"] = "not written: %s" % e
        print(compact_line(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


LINE_LIMIT = 4096   # bytes; the driver keeps the last 8 KB of stdout (BENCH_r02.json: a 59 KB line, parsed: null)


def compact_line(out, limit=LINE_LIMIT):
    """The ONE stdout line: json of `out`, with the optional detail dropped (most dispensable first)
    until it fits `limit` bytes.  The contract's keys, roofline's contract fields and cpu_baseline
    are never dropped."""
    out = json.loads(json.dumps(out))  # deep copy
    drops = [("roofline", "kernel_ms_per_pattern"), ("roofline", "traffic_source"), ("config", "prewarm"),
             ("config", "corpus"), ("config", "sharding"), ("cpu_baseline", "all_cores"), ("rccl", "devices"), ("roofline_secondary", "cells"), ("own_kernel_min",),
             ("roofline", "kernels_of_the_timed_plans"), ("counts_verified",), ("worst_cells",), ("min_frac",)]
    line = json.dumps(out, separators=(",", ":"))
    for path in drops:
        if len(line) <= limit:
            break
        d = out
        for k in path[:-1]:
            d = d.get(k) or {}
        d.pop(path[-1], None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > limit:
        raise SystemExit("bench line is %d bytes (> %d) even without its optional detail" % (len(line), limit))
    return line


# the configurations of BASELINE.json that NAME an algorithm (config 2: HOR; 3: SO, BNDM; 4: BM; 5: HOR BM KMP SO EPSM)
NAMED_IN = {"hor": (5,), "bm": (4, 5), "kmp": (5,), "so": (3, 5), "bndm": (3,), "epsm": (5,)}


def own_kernel_summary(cells):
    """Per algorithm: the worst cell measured ON THE ALGORITHM'S OWN KERNEL (a cell the plan did not
    reroute, or the cell's second entry under smartgpu_tune(0,1)) on rand128 m in {4..256} and on
    each configuration that names the algorithm: [frac, "sigma/m", worst frac among the cells with
    m >= 16] — below 16 bytes a skip loop visits a window every byte or two and is iteration-bound
    on any text."""
    out = {}
    for algo, own in OWN_KERNEL.items():
        mine = [c for c in cells if c["algo"] == algo and c["kernel"] == own and "kernels" not in c]
        entry = {}
        for label, sel in [("rand128", lambda c: c["config"] == 2)] + [("config%d" % k, (lambda c, k=k: c["config"] == k)) for k in NAMED_IN[algo]]:
            got = [c for c in mine if sel(c)]
            if got:
                w = min(got, key=lambda c: c["frac"])
                long_ = [c["frac"] for c in got if c["m"] >= 16]
                entry[label] = [w["frac"], "%s/m%d" % (w["sigma"], w["m"]), min(long_) if long_ else None]
        out[algo] = entry
    return out


def runs_data_path_summary(cells):
    """The second roofline of the runs kernels: their loader's DATA PATH.  kmp_runs and so_runs move the text the same
    way — whole 128-byte lines of 64 runs per wave through 4 KB LDS slabs — and that path alone (automaton compiled out)
    runs at 78-80 % of 8 TB/s; so_runs sits within 2 % of it on every input, so its time on the same
    text and pattern length, measured in the same sweep, is the live stand-in for that ceiling.  Reported for kmp_runs —
    the kernel furthest below the HBM roofline among the plans' choices — on the configuration-2 cells both ran:
    achieved / peak in GB/s, frac = kmp_runs' share of the data path."""
    pairs = []
    for c in cells:
        if c["config"] == 2 and c["algo"] == "kmp" and c["kernel"] == "kmp_runs" and not c.get("own_kernel"):
            so = [x for x in cells if x["config"] == 2 and x["algo"] == "so" and x["m"] == c["m"] and x["kernel"] == "so_runs"]
            if so:
                pairs.append((c["m"], c["frac"], so[0]["frac"]))
    if not pairs:
        return None
    m, kmp, so = min(pairs, key=lambda t: t[1] / t[2])
    return {"kernel": "kmp_runs", "bound": "runs loader data path (so_runs, same text and m, same sweep)", "workload": "rand128 1 GiB m=%d" % m,
            "achieved": round(kmp * HBM_PEAK_GBS, 1), "peak": round(so * HBM_PEAK_GBS, 1), "unit": "GB/s", "frac": round(kmp / so, 4),
            "cells": [[mm, k, s_] for mm, k, s_ in pairs]}


def worst_cells_summary(cells, k=3):
    """Per BASELINE configuration the k worst cells under the plan's kernel choice and the k worst measured on the
    algorithm's OWN kernel (cells the plan did not reroute, and the second entries of rerouted ones), each as
    [algo, sigma, m, kernel, frac] — so that the driver's record of the line says WHICH cells set min_frac and
    own_kernel_min, not only their values (the cells themselves are in the sweep file)."""
    out = {}
    for cfg in sorted({c["config"] for c in cells}):
        mine = [c for c in cells if c["config"] == cfg]
        plan = sorted((c for c in mine if not c.get("own_kernel")), key=lambda c: c["frac"])[:k]
        own = sorted((c for c in mine if c["kernel"] == OWN_KERNEL.get(c["algo"]) and "kernels" not in c), key=lambda c: c["frac"])[:k]
        row = lambda c: [c["algo"], c["sigma"], c["m"], c["kernel"].replace("_scan", "").replace("_runs", ""), c["frac"]]  # noqa: E731
        out["config%d" % cfg] = {"plan": [row(c) for c in plan], "own": [row(c) for c in own]}
    return out


def run_sweep(text128, device):
    """Cells of BASELINE config 2 (six algorithms x m in {4,8,32,256}, 1 GiB rand128), config 3
    (SO and BNDM x sigma in {2,4} x m in {2,4,8,16,32,64}, 1 GiB), config 4 (the English unit
    bible.txt||world192.txt tiled to its stated 4 GiB, six algorithms x the lengths of sets.h:25) and
    config 5's alphabets (sigma in {2,32,256} x HOR/BM/KMP/SO/EPSM x sets.h:25, the 4 GiB shard one of its
    eight GPUs holds — round 4: these two ran on 1 GiB texts before; BIG_GIB below):
    the harness loop of src/smart.c:290-345 reduced to what it times: per cell 4 rounds of 3 patterns,
    each round between two HIP events on the launch stream; a cell's time is its median round.  Every cell names the kernel its plans
    launched; a cell whose plans were rerouted (api.cpp build_blob) is measured again on the
    algorithm's own kernel."""
    import numpy as np
    import smart_amd
    from smart_amd import Plan, Text, engine

    n = 1 << 30
    BIG = int(BIG_GIB * (1 << 30))  # configs 4 and 5: the size BASELINE.json states per GPU
    J, REPS = 3, 4
    cells = []
    ref_counts = {}
    checked_by = collections.Counter()

    def time_cell(config, text, sigma, algo, m, pats, own, n):
        if own:
            engine.tune(0, 1)
        try:
            plans = [Plan(algo, p, device=device) for p in pats]
            kernels = collections.Counter(pl.kernel_name for pl in plans)
            plans[0].launch(text, slot=1)  # warm-up (code object, LDS attribute)
            engine.device_sync(device)
            rounds = []
            for r in range(REPS):  # every round between its own pair of events: ONE stall of the box inside a single
                engine.stream_mark(device, 0)  # 12-launch interval read as a 58 % cell once (hor m=256, 84 % in every sweep)
                for pl in plans:
                    pl.launch(text, slot=0)
                engine.stream_mark(device, 1)
                rounds.append(engine.stream_elapsed_ms(device) / len(plans))
            rounds.sort()
            ms = 0.5 * (rounds[(REPS - 1) // 2] + rounds[REPS // 2])  # the median round
            counts = [pl.result(0)[0] // REPS for pl in plans]
        finally:
            if own:
                engine.tune(0, 0)
        # the reference count: a kernel of ANOTHER family than the one that produced the count — the first of the
        # packed matcher, the KMP automaton and Shift-Or whose plan for THIS pattern launches a different kernel, under
        # the plan's choice or, failing that, on its own kernel (on sigma 2/4 at m = 8 the plans of EPSM, KMP and SO all
        # count on so_runs: comparing so_runs with itself checks nothing, packed_scan under tune(0,1) does)
        ok = True
        for j, (p, pl) in enumerate(zip(pats, plans)):
            fam = None
            for tuned in (0, 1):
                engine.tune(0, tuned)
                try:
                    fam = next((f for f in ("epsm", "kmp", "so") if engine.kernel_for(f, p) != pl.kernel_name), None)
                    if fam is not None:
                        key = (sigma, n, m, j, fam, tuned)
                        if key not in ref_counts:
                            ref_counts[key] = smart_amd.search(fam, p, text)[0]
                        checked_by[engine.kernel_for(fam, p)] += 1
                finally:
                    engine.tune(0, 0)
                if fam is not None:
                    break
            assert fam is not None, (algo, m, sigma, pl.kernel_name)
            ok = ok and counts[j] == ref_counts[key] and counts[j] >= 1
        for pl in plans:
            pl.free()
        kernel = kernels.most_common(1)[0][0]
        cell = {"config": config, "algo": algo, "m": m, "sigma": sigma, "kernel": kernel, "ms": round(ms, 4), "gib": round(n / (1 << 30), 3),
                "frac": round(n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "count_ok": bool(ok)}
        if len(kernels) > 1:
            cell["kernels"] = dict(kernels)
        if own:
            cell["own_kernel"] = True  # smartgpu_tune(0,1): the algorithm's own kernel, not the plan's choice
        return cell

    def cells_for(config, text, sigma, algos, ms, unit=None, n=n):
        for m in ms:
            if unit is None:
                pats = [text.pattern(splitmix64(PATTERN_SALT + 4096 * j + m) % (n - m), m) for j in range(J)]
            else:  # config 4: patterns from the first copy of the unit
                pats = [unit[k:k + m] for k in (splitmix64(PATTERN_SALT + 4096 * j + m) % (len(unit) - m) for j in range(J))]
            for algo in algos:
                if m < smart_amd.MIN_M.get(algo, 1):
                    continue
                c = time_cell(config, text, sigma, algo, m, pats, False, n)
                cells.append(c)
                if c["kernel"] != OWN_KERNEL[algo] or "kernels" in c:
                    cells.append(time_cell(config, text, sigma, algo, m, pats, True, n))

    own128 = text128 is None
    if own128:
        text128 = Text.generate(SEED, 128, n, device=device)
    SETS_H_25 = (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)  # src/sets.h:25
    cells_for(2, text128, 128, ("hor", "bm", "kmp", "so", "bndm", "epsm"), (4, 8, 32, 256))
    if own128:
        text128.free()
    for sigma in (4, 2):
        t = Text.generate(SEED, sigma, n, device=device)
        cells_for(3, t, sigma, ("so", "bndm"), (2, 4, 8, 16, 32, 64))
        t.free()
    for sigma in (2, 32, 256):
        t = Text.generate(SEED, sigma, BIG, device=device)
        cells_for(5, t, sigma, ("hor", "bm", "kmp", "so", "epsm"), SETS_H_25, n=BIG)
        t.free()
    from smart_amd import corpus
    unit = corpus.english_unit()
    t = Text.upload_tiled(unit, BIG, device=device)
    cells_for(4, t, "english", ("hor", "bm", "kmp", "so", "bndm", "epsm"), SETS_H_25, unit=unit, n=BIG)
    t.free()
    bad = [c for c in cells if not c["count_ok"]]
    if bad:
        raise SystemExit("SWEEP COUNT MISMATCH: %s" % bad)
    north = [c["frac"] for c in cells if c["config"] == 2 and not c.get("own_kernel")]
    plan = lambda k: [c["frac"] for c in cells if c["config"] == k and not c.get("own_kernel")]  # noqa: E731
    return {"cells": cells, "own_kernel_min": own_kernel_summary(cells), "worst_cells": worst_cells_summary(cells),
            "roofline_secondary": runs_data_path_summary(cells),
            "min_frac": {"rand128_m4to256_plan_choice": min(north),
                         "config3_plan_choice": min(plan(3)), "config4_english_plan_choice": min(plan(4)),
                         "config5_plan_choice": min(plan(5)),
                         "plan_choice_all_cells": min(c["frac"] for c in cells if not c.get("own_kernel")),
                         # what the plans route AWAY from: a skip loop on a binary text shifts by a byte or two
                         "own_kernel_cells": min([c["frac"] for c in cells if c.get("own_kernel")] or [None])},
            "note": "config = BASELINE.json configuration the cell belongs to (2, 3: 1 GiB texts; 4: the English unit tiled to %g GiB; 5: one "
                    "%g GiB shard per alphabet — the sizes BASELINE.json states per GPU; 32 GiB on one GPU: tests/test_configs_gpu.py); "
                    "gib = the cell's text size; ms = the median of %d rounds, each %d patterns between two HIP events; frac = text bytes / ms / 8 TB/s; "
                    "own_kernel = measured again with smartgpu_tune(0,1) because the plan rerouted the pattern; "
                    "count_ok = equal to the count of a kernel of another family (reference kernels used: %s)"
                    % (BIG_GIB, BIG_GIB, REPS, J, dict(checked_by))}


if __name__ == "__main__":
    main()
