"""World-size-2 test of the sharded path on CPU (gloo): the shard arithmetic of
smart_amd.sharding plus the single collective (a sum of counts).  Each rank
counts its shard with the CPU oracle — the GPU kernels are covered by
test_parity_gpu.py; what is checked here is that ownership by start offset with
an (m-1)-byte overlap neither loses nor double-counts occurrences."""
import os
import socket

import numpy as np
import pytest

from smart_amd.sharding import split_starts, weak_shard


def test_split_starts_partitions_every_start():
    for n in (0, 1, 5, 64, 1000, 12345):
        for m in (1, 2, 7, 64, 999, 2000):
            for world in (1, 2, 3, 8):
                covered = []
                for g in range(world):
                    off, ln = split_starts(n, m, g, world)
                    if ln >= m:
                        covered.extend(range(off, off + ln - m + 1))
                        assert off + ln <= n
                want = list(range(0, n - m + 1)) if n >= m else []
                assert covered == want, (n, m, world)


def test_weak_shard_covers_global_text():
    S, m, world = 1000, 33, 4
    total = S * world
    starts = []
    for g in range(world):
        off, ln = weak_shard(S, m, g, world)
        assert off + ln <= total
        starts.extend(range(off, off + ln - m + 1))
    assert starts == list(range(total - m + 1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        results = []
        for sigma, n, m, k in ((2, 300000, 5, 777), (4, 200001, 31, 1), (128, 100000, 64, 99000), (2, 50000, 1, 0)):
            T = pyoracle.gen_text(1234, sigma, 0, n)
            P = T[k:k + m]
            # strong split of one text
            off, ln = split_starts(n, m, rank, world)
            local = pyoracle.search("hor", P, T[off:off + ln]) if ln >= m else 0
            t = torch.tensor([local], dtype=torch.int64)
            dist.all_reduce(t)  # the path's only collective: a sum of counts
            whole = pyoracle.search("hor", P, T)
            results.append((int(t[0]), whole))
            # weak shards of a world*S text generated piecewise by offset
            S = n // world
            goff, glen = weak_shard(S, m, rank, world)
            piece = pyoracle.gen_text(1234, sigma, goff, glen)
            assert np.array_equal(piece, T[goff:goff + glen])
            t = torch.tensor([pyoracle.search("bm", P, piece)], dtype=torch.int64)
            dist.all_reduce(t)
            results.append((int(t[0]), pyoracle.search("bm", P, T[:S * world])))
        if rank == 0:
            out.put(results)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_count_equals_whole(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(results) == 8
    for got, want in results:
        assert got == want
