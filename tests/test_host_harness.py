"""The C harness (smart_amd/host): argument handling on CPU, full runs on GPU."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "smart_amd", "bin")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "smart_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "smart_amd", "host")])


def run(*args, cwd=None, env=None):
    return subprocess.run([os.path.join(BIN, args[0])] + list(args[1:]), capture_output=True, text=True,
                          cwd=cwd or ROOT, timeout=600, env=dict(os.environ, **env) if env else None)


def test_smart_argument_errors_match_reference_messages():
    # messages of src/smart.c:438-548
    assert "No parameter given. Use -h for help." in run("smart").stdout
    assert "-pset N" in run("smart", "-h").stdout
    assert "Error in input parameters. Use -h for help." in run("smart", "-bogus").stdout
    assert "Error in input parameters. Use -h for help." in run("smart", "-pset", "abc").stdout
    assert "The minimum length is not a valid argument" in run("smart", "-plen", "0", "5").stdout
    assert "The maximum length is not a valid argument" in run("smart", "-plen", "9", "5").stdout
    assert "Both parameters -simple and -text defined" in run("smart", "-simple", "ab", "abab", "-text", "rand2").stdout
    assert "No filename given" in run("smart", "-occ").stdout
    assert "Unknown algorithm" in run("smart", "-algo", "nope", "-text", "rand2").stdout


def test_textgen_reproduces_the_reference_corpora(tmp_path, oracle):
    """smart_amd/bin/textgen restates src/textgen.c:34-54 without libc's rand(): the eight corpora
    have the md5 sums of the reference's files and equal the oracle's stream."""
    import hashlib
    import numpy as np
    r = run("textgen", "-data", str(tmp_path / "data"))
    assert r.returncode == 0, r.stdout + r.stderr
    md5 = {2: "a8e4cecf43689f3964b4fdfe191aa01b", 4: "c371ad58eda30fa18f35a77b1e775ece",
           8: "84f8e24bad0787e2caaa250cd0ca715a", 16: "f288f883f5ee611b7667bf25dba9f4c0",
           32: "743a9837b8172b9e408617b675b97b57", 64: "a89762a4005f7a8c4885d47dcc5da8a1",
           128: "44ef38efacf46e3705d14f1ed4399faf", 250: "8cf7fb1486e0578f596aa1ac441be246"}
    for sigma, want in md5.items():
        d = tmp_path / "data" / ("rand%d" % sigma)
        blob = (d / ("rand%d.txt" % sigma)).read_bytes()
        assert len(blob) == 5000000 and hashlib.md5(blob).hexdigest() == want, sigma
        assert ("#rand%d.txt#" % sigma) in (d / "index.txt").read_text()
    assert np.array_equal(np.frombuffer(blob[:100000], dtype=np.uint8), oracle.textgen(250, 100000))


def test_test_tool_usage():
    assert "usage: ./test ALGONAME" in run("test").stdout
    assert run("test", "nope").returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita", "hash3", "hash5", "hash8", "sbndm", "kr", "bndml"])
def test_reference_cases_through_plugin_shape(algo):
    r = run("test", algo)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Well done! Test passed successfully" in r.stdout


@pytest.mark.gpu
def test_smart_report_lines(tmp_path):
    r = run("smart", "-text", "rand128", "-plen", "32", "32", "-pset", "5", "-occ", "-pre", "-dif", "-std", "-txt",
            "-tex", "-php", "-tb", "60000", cwd=str(tmp_path))  # -tb: a hiccup of a shared box must not turn a row into [OUT]
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "Searching for a set of 5 patterns with length 32" in out
    assert "Testing 16 algorithms" in out
    for name in ("HOR", "BM", "KMP", "SO", "BNDM", "EPSM", "SA", "QS", "TUNEDBM", "RAITA", "HASH3", "HASH5", "HASH8", "SBNDM", "KR", "BNDML"):
        line = [ln for ln in out.splitlines() if re.search(r"\] %s \." % name, ln)]
        assert line and "[OK]" in line[0] and "occ 1" in line[0] and "GB/s" in line[0], (name, out)
        assert re.search(r"\d+\.\d\d+ \+ \d+\.\d\d+ ms", line[0])  # %.2f as the reference above 1 ms, %.4f below
        # GB/s, its share of the HBM-read roofline, the GPUs, the kernel that ran (SURVEY.md §5 metrics row)
        assert re.search(r"\d+\.\d GB/s\t\d+\.\d% of 1 x 8 TB/s\t(hor_scan|bm_scan|kmp_runs|so_runs|bndm_scan|sbndm_scan|packed_scan|hor_scan_bp|bndml_scan)$", line[0]), line[0]
    table = list((tmp_path / "results").glob("EXP*/rand128.txt"))
    assert table and table[0].read_text().startswith("HOR")
    xml = list((tmp_path / "results").glob("EXP*/rand128.xml"))[0].read_text()
    assert xml.startswith("<RESULTS>") and xml.count("<NAME>") == 16 and "<SEARCH>" in xml and "<BEST>" in xml
    assert xml.count("<KERNEL>") == 16 and "<KERNEL>kmp_runs</KERNEL>" in xml
    assert "<GPUS>1</GPUS>" in xml and "<HBMPEAK" in xml and xml.count("<ROOFLINE>") == 16
    gbs, roof = float(re.search(r"<GBS>([\d.]+)</GBS>", xml).group(1)), float(re.search(r"<ROOFLINE>([\d.]+)</ROOFLINE>", xml).group(1))
    assert abs(roof - gbs / 8000.0) < 2e-4
    roofs = list((tmp_path / "results").glob("EXP*/rand128.roofline.txt"))[0].read_text().splitlines()
    assert roofs[0].startswith("# GPUs 1\tHBM peak 8000 GB/s per GPU") and len(roofs) == 17
    assert re.fullmatch(r"HOR +\t\d+\.\d% \(\d+\.\d\)", roofs[1]), roofs[1]
    kernels = list((tmp_path / "results").glob("EXP*/rand128.kernels.txt"))[0].read_text().splitlines()
    assert len(kernels) == 16 and kernels[0].split() == ["HOR", "hor_scan"] and kernels[2].split() == ["KMP", "kmp_runs"]
    html = list((tmp_path / "results").glob("EXP*/rand128.html"))[0].read_text()
    assert html.startswith("<!DOCTYPE html>") and html.count("<tr><td class=\"algo\">") == 16 and "class=\"best\"" in html
    assert html.count("<svg ") == 2 and html.count("<polyline ") == 32  # the two charts: a line per algorithm each
    php = list((tmp_path / "results").glob("EXP*/rand128.php"))[0].read_text()  # outputPHP's array (output.h:49-113)
    assert php.startswith("<?\n$rand128 = array(\n\t\"PATT\" => array(\"32\", ),") and php.endswith(");\n?>")
    assert re.search(r'"HOR" => array\("\d+\.\d{4}", \),', php) and '"KMP.best" => array(' in php and '"BNDML.std" => array(' in php
    index = list((tmp_path / "results").glob("EXP*/index.html"))[0].read_text()  # outputINDEX (output.h:706-741)
    assert '<a href="rand128.html">Experimental results on rand128</a>' in index
    tex = list((tmp_path / "results").glob("EXP*/rand128.tex"))[0].read_text()
    assert tex.startswith("\\begin{tabular}{|l|l|}") and "\\textsc{HOR} & " in tex and tex.endswith("\\end{tabular}")
    # -simple: the reference's own example (SURVEY.md §5 hazard 3 segfaults EPSM there)
    r = run("smart", "-simple", "aba", "ababababab", "-pset", "1", "-occ", cwd=str(tmp_path))
    assert r.stdout.count("occ 4") == 14, r.stdout  # hash5/hash8 do not apply to m = 3


@pytest.mark.gpu
def test_smart_gpus_rehearsal_on_one_box(tmp_path):
    """`smart -gpus 3` (smart.c:140-146's call replaced by the sharded smartgpu_msearch_batch64) on a box
    with fewer GPUs: SMARTGPU_REDUCE_HOST=1 puts shard g on GPU g mod visible and adds the counts on the
    host.  The harness path, its shard arithmetic and its report columns run; the times mean nothing."""
    r = run("smart", "-text", "rand4", "-plen", "2", "64", "-pset", "4", "-occ", "-seed", "7", "-tb", "60000", "-gpus", "3", "-algo", "hor,bm,kmp,so,bndm,epsm",
            cwd=str(tmp_path), env={"SMARTGPU_REDUCE_HOST": "1"})
    single = tmp_path / "single"  # its own directory: both runs may get the same EXP<time> code and write the same files
    single.mkdir()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Text sharded over 3 GPUs" in r.stdout and "on the host (SMARTGPU_REDUCE_HOST: rehearsal)" in r.stdout
    one = run("smart", "-text", "rand4", "-plen", "2", "64", "-pset", "4", "-occ", "-seed", "7", "-tb", "60000", "-algo", "hor,bm,kmp,so,bndm,epsm", cwd=str(single))
    assert one.returncode == 0, one.stdout + one.stderr
    rows = [ln for ln in r.stdout.splitlines() if "[OK]" in ln]
    assert len(rows) == 6 * 6 and all("% of 3 x 8 TB/s" in ln for ln in rows), r.stdout
    assert "[ERROR]" not in r.stdout and "[--]" not in r.stdout
    # same seed, same generated corpus: the same patterns, so the sharded run reports the very same occurrence means
    occ = lambda out: re.findall(r"\] (\w+) \..*\[OK\].*occ (\d+)", out)  # noqa: E731
    # (-tb 60000: the default 300 ms limit turns a row into [OUT] when a shared box hiccups — seen once in five runs)
    assert occ(r.stdout) == occ(one.stdout) and len(occ(one.stdout)) == 36, (r.stdout, one.stdout)
    xml = list((tmp_path / "results").glob("EXP*/rand4.xml"))
    assert any("<GPUS>3</GPUS>" in x.read_text() for x in xml)
    # without the rehearsal switch more GPUs than the box has is an error, not a silent fallback
    bad = run("smart", "-text", "rand4", "-plen", "2", "2", "-pset", "1", "-gpus", "64", cwd=str(tmp_path))
    assert bad.returncode == 1 and "only" in bad.stderr


@pytest.mark.gpu
def test_smart_times_every_pattern_when_spread_or_bound_are_asked_for(tmp_path, oracle):
    """src/smart.c:320-329 times every pattern, :337-343 applies -tb per run, :347-351 derives best / worst / std from
    those times.  With -dif, -std or -tb the harness launches and times every pattern on its own
    (smartgpu_search_batch64_each) — so ONE deliberately slow pattern in a set shows as the worst time, gives a standard
    deviation, and trips [OUT] when the bound lies between the fast and the slow ones; without those flags the set
    shares one grid (texts up to 32 MiB) and only the mean is reported.
    The corpus: 8 MiB of rand128 followed by 8 MiB of 'a' — a pattern cut from the second half is a candidate at
    every position of that half for the packed matcher (EPSM verifies each of them in memory), one from the first half
    is a candidate nowhere else."""
    import numpy as np
    d = tmp_path / "data" / "rand128"
    d.mkdir(parents=True)
    half = 8 << 20
    body = np.concatenate([oracle.gen_text(4242, 128, 0, half) + 1, np.full(half, ord("a"), np.uint8)])  # (+1: no NUL bytes)
    (d / "slow.txt").write_bytes(body.tobytes())
    (d / "index.txt").write_text("#slow.txt#\n")
    common = ("smart", "-text", "rand128", "-data", str(tmp_path / "data"), "-tsize", "16", "-plen", "32", "32", "-pset", "12",
              "-seed", "11", "-algo", "epsm", "-occ")
    r = run(*common, "-dif", "-std", "-tb", "60000", cwd=str(tmp_path))
    assert r.returncode == 0 and "[OK]" in r.stdout, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if "] EPSM ." in ln][0]
    best, worst = (float(x) for x in re.search(r"\[(\d+\.\d+), (\d+\.\d+)\]", line).groups())
    std = float(re.search(r"std (\d+\.\d+)", line).group(1))
    assert worst > 4 * best and std > 0, line  # the slow patterns stand out: every pattern was timed on its own
    # a bound between the fast and the slow patterns: the first slow one ends the algorithm with [OUT] (smart.c:337-343)
    bound = "%.4f" % ((best * worst) ** 0.5)
    out = run(*common, "-tb", bound, cwd=str(tmp_path))
    assert out.returncode == 0 and "[OUT]" in out.stdout and "[OK]" not in out.stdout, (bound, out.stdout)
    # a bound below the fastest: [OUT] as well; far above the slowest: [OK]
    assert "[OUT]" in run(*common, "-tb", "%.5f" % (best / 4), cwd=str(tmp_path)).stdout
    # without the flags: the set shares a grid, the mean is reported
    plain = run(*common, cwd=str(tmp_path))
    assert plain.returncode == 0 and "[OK]" in plain.stdout and "std" not in plain.stdout, plain.stdout
