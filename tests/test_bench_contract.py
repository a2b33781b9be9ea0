"""The bench line's contract, checked on the CPU against the lines committed under profiles/
(each was printed by `python bench.py` on an MI355X box and copied by tools/collect_profiles.py),
and bench.py's helpers that need no GPU."""
import glob
import json
import os

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


def test_splitmix_and_traffic_table():
    assert bench.splitmix64(0) == 0xE220A8397B1DCDAF  # the published first output of SplitMix64
    t = bench.load_traffic("hor_scan", "hor_m32_sigma128_gib1")
    assert t is not None and 1.0 <= t / 2**30 < 1.05
    assert bench.load_traffic("hor_scan", "no_such_workload") is None
