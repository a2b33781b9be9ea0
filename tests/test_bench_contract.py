"""The bench line's contract, checked on the CPU against the lines committed under profiles/
(each was printed by `python bench.py` on an MI355X box and copied by tools/collect_profiles.py),
and bench.py's helpers that need no GPU."""
import glob
import json
import os

import pytest

from conftest import ROOT

import bench

REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict, "cpu_baseline": dict}


def test_committed_bench_lines_keep_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_bench_default.json")))
    assert files, "no bench line under profiles/"
    for f in files:
        line = open(f).read().strip()
        assert "\n" not in line, f  # ONE JSON line
        d = json.loads(line)
        for k, t in REQUIRED.items():
            assert isinstance(d.get(k), t), (f, k)
        assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no MI355X number for this metric
        assert d["unit"] == "GB/s" and d["dtype"] == "u8" and d["scaling"] == "weak" and d["higher_is_better"] is True
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is None or r["traffic"] >= r["bytes_per_launch"]  # HBM bytes by PMC >= algorithmic bytes
        # whole-job value and the kernel-only rate differ by launch gaps and the read-back only
        assert 0.9 < d["value"] / (d["n_gpus"] * r["achieved"]) <= 1.0001
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == "GB/s" and c["value"] > 0 and c["sample"]


def test_bench_lines_fit_the_drivers_stdout_tail():
    """The driver keeps 8 KB of stdout: round 2's line carried its 433 sweep cells (59 KB) and was cut off
    (BENCH_r02.json: parsed null).  From round 3 on a committed line is at most 4 KB and the cells live in a file."""
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_bench_default.json")))
             if int(os.path.basename(os.path.dirname(f))[1:]) >= 3]
    for f in files:
        line = open(f).read().strip()
        assert len(line) <= bench.LINE_LIMIT, (f, len(line))
        d = json.loads(line)
        assert "sweep" not in d and isinstance(d.get("min_frac"), dict) and len(d.get("own_kernel_min", {})) <= 12


def test_compact_line_keeps_the_contract_under_4k():
    cells = [{"config": c, "algo": a, "m": m, "sigma": s, "kernel": bench.OWN_KERNEL[a] if own else "so_runs", "ms": 0.2,
              "frac": 0.5 + 0.001 * m ** 0.5, "count_ok": True, **({"own_kernel": True} if own else {})}
             for c, s in ((2, 128), (3, 4), (3, 2), (4, "english"), (5, 2), (5, 32), (5, 256))
             for a in bench.OWN_KERNEL for m in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096) for own in (False, True)]
    own = bench.own_kernel_summary(cells)
    assert set(own) == set(bench.OWN_KERNEL) and own["bm"].keys() == {"rand128", "config4", "config5"}
    assert own["bndm"]["config3"][1] in ("4/m2", "2/m2") and own["bndm"]["config3"][2] >= own["bndm"]["config3"][0]
    out = {"metric": "x", "value": 1.0, "unit": "GB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.1,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "w", "corpus": "c" * 300, "prewarm": "p" * 300, "sharding": "s" * 300},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None,
                        "traffic_source": "t" * 900, "kernel_ms_per_pattern": {"note": "n" * 900}},
           "cpu_baseline": {"value": 1.0, "unit": "GB/s", "cores": 1, "kind": "reference", "sample": "s", "all_cores": {"x": "y" * 900}},
           "counts_verified": "v" * 300, "min_frac": {"a": 0.5}, "own_kernel_min": own}
    line = bench.compact_line(out)
    assert len(line) <= bench.LINE_LIMIT and "\n" not in line
    d = json.loads(line)
    for k, t in REQUIRED.items():
        assert isinstance(d.get(k), t), k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and "workload" in d["config"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    small = dict(out, config={"workload": "w"})
    assert json.loads(bench.compact_line(small)) == json.loads(json.dumps(small)) or len(json.dumps(small)) > bench.LINE_LIMIT


def test_latest_bench_line_agrees_with_its_rocprof_summary():
    """roofline.kernel_ms (HIP events inside bench.py) against the rocprofv3 --kernel-trace --stats
    summary of the same command, committed next to it."""
    import csv
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_bench_default.json")))  # session letters sort by age
    d = json.loads(open(lines[-1]).read())
    stats = lines[-1].replace("_bench_default.json", "_bench_hor_m32_kernel_stats.csv")
    assert os.path.exists(stats), stats
    rows = [r for r in csv.DictReader(open(stats)) if d["roofline"]["kernel"] in r["Name"]]
    assert rows, "the roofline kernel is not in the rocprof summary"
    avg_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - d["roofline"]["kernel_ms"]) / d["roofline"]["kernel_ms"] < 0.08, (avg_ms, d["roofline"]["kernel_ms"])


def test_splitmix_and_traffic_table(tmp_path):
    """roofline.traffic is bound to the kernel source it was profiled on: the table records, per kernel, the sha256
    of that kernel FAMILY's sources (one translation unit per family: smart_amd/sources.py), and bench.py reports
    null (and why) for any other source — while an edit to another family's unit leaves the figure valid."""
    from smart_amd import sources
    assert bench.splitmix64(0) == 0xE220A8397B1DCDAF  # the published first output of SplitMix64
    table = {"_source": {"commit": "abc1234", "summary": "profiles/rXX/x.csv"},
             "hor_scan": {"hor_m32_sigma128_gib1": 1082204320, "_sha256": bench.kernel_sha256("hor_scan")},
             "kmp_runs": {"kmp_m32_sigma128_gib1": 1101216064, "_sha256": "0" * 64}}  # profiled on another kmp_runs
    f = tmp_path / "pmc_traffic.json"
    f.write_text(json.dumps(table))
    t, src = bench.load_traffic("hor_scan", "hor_m32_sigma128_gib1", path=str(f))
    assert t == 1082204320 and "abc1234" in src and "profiles/rXX/x.csv" in src
    t, why = bench.load_traffic("hor_scan", "no_such_workload", path=str(f))
    assert t is None and "no PMC pass" in why
    t, why = bench.load_traffic("kmp_runs", "kmp_m32_sigma128_gib1", path=str(f))
    assert t is None and "not measured for this kernel source" in why
    t, why = bench.load_traffic("bm_scan", "bm_m32_sigma128_gib1", path=str(f))
    assert t is None and "no PMC pass" in why
    # the families' source sets are disjoint apart from the common headers: every unit's own file is in one set only
    own = [f for u in sources.UNITS.values() for f in u if f.startswith("k_")]
    assert len(own) == len(set(own)) == len(sources.UNITS)
    assert all(os.path.exists(p) for u in sources.UNITS for p in sources.unit_files(u))
    assert sources.kernel_sha256("hor_scan") != sources.kernel_sha256("kmp_runs")
    # the committed table: every kernel's entry either matches its family's committed sources, or bench.py says null
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for kernel, entry in committed.items():
        if kernel.startswith("_"):
            continue
        for key, v in entry.items():
            if key == "_sha256":
                continue
            t, src = bench.load_traffic(kernel, key)
            if entry["_sha256"] == bench.kernel_sha256(kernel):
                assert t == v and 1.0 <= t / 2**30 < 1.05, (kernel, key, t)
            else:
                assert t is None and src


def test_plain_multi_gpu_invocation_launches_its_ranks():
    """`python bench.py --gpus 2` without torchrun: the process starts the two ranks itself (before it
    touches a GPU) through torch.distributed.run on 127.0.0.1; --check-launch makes the ranks only
    rendezvous (gloo), all-reduce their rank ids and report — no GPU needed."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"check_launch": True, "n_gpus": 2, "rank_id_sum": 3, "launched_by": "torch.distributed.run",
                 "master_addr": "127.0.0.1"}
    # the node's eight ranks (gloo, no GPU): the launcher starts all of them and they all meet
    out8 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--check-launch"],
                          env=env, capture_output=True, text=True, timeout=600)
    assert out8.returncode == 0, out8.stderr[-2000:]
    d8 = json.loads([ln for ln in out8.stdout.splitlines() if ln.startswith("{")][-1])
    assert d8["n_gpus"] == 8 and d8["rank_id_sum"] == 36 and d8["master_addr"] == "127.0.0.1"
    # a rank count that does not match the launcher's world size is refused
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                         env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in (bad.stderr + bad.stdout)


@pytest.mark.gpu
def test_two_ranks_rehearsed_on_one_gpu():
    """bench.py's multi-rank code on a one-GPU box: two gloo ranks share GPU 0, each holds its shard of a 2 x 64 MiB
    text with the (m-1)-byte overlap, the counts are summed with one all-reduce.  The numbers mean nothing; the line,
    the rank plumbing (torch.distributed.run on 127.0.0.1, started by bench.py itself) and the counts are checked —
    bench.py aborts on any count that differs from a kernel of another family."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                          "--steps", "5", "--warmup", "2", "--gib", "0.0625", "--no-sweep", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= bench.LINE_LIMIT  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["ranks"] == 2 and "gloo, rehearsal" in d["config"]["sharding"]
    # the live group: gloo, two ranks — both on the one GPU of this box, and the record says so
    assert d["rccl"]["world_size"] == 2 and d["rccl"]["backend"] == "gloo" and d["rccl"]["distinct_devices"] == 1
    assert [r[0] for r in d["rccl"]["devices"]] == [0, 1]
    assert abs(d["value_per_gpu"] * 2 - d["value"]) < 0.02 and d["aggregate"] == d["value"]
    assert d["config"]["text_bytes_per_gpu"] == (1 << 26) + 31  # rank 0's shard: its starts plus m-1 bytes of the next
    assert "all 5 counts equal" in d["counts_verified"] and d["roofline"]["bytes_per_launch"] == (1 << 26) + 31
