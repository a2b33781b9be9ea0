"""The registry editor `select` (smart_amd/host/select.c) against what src/select.c:57-194 does with
source/algorithms.h — the file `smart` reads when no -a list is given (getAlgo, src/function.h:62-77).  No GPU: the
engine enters only through `./test NAME -nv`, which these tests replace by a script."""
import os
import stat
import subprocess

from conftest import ROOT

import pytest

SELECT = os.path.join(ROOT, "smart_amd", "bin", "select")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "smart_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "smart_amd", "host")])


def run(cwd, *args):
    r = subprocess.run([SELECT, *args], cwd=cwd, capture_output=True, text=True, timeout=60)
    return r.returncode, r.stdout


def registry(cwd):
    return open(os.path.join(cwd, "source", "algorithms.h")).read()


def make_tree(tmp_path, lines, test_status=0):
    (tmp_path / "source" / "bin").mkdir(parents=True)
    (tmp_path / "source" / "algorithms.h").write_text("".join("#%d #%s \n" % (f, n) for f, n in lines))
    t = tmp_path / "test"
    t.write_text("#!/bin/sh\necho \"$@\" >> tested\nexit %d\n" % test_status)
    t.chmod(t.stat().st_mode | stat.S_IEXEC)
    return str(tmp_path)


def test_binary_is_built():
    assert os.access(SELECT, os.X_OK), "make -C smart_amd/host builds smart_amd/bin/select"


def test_toggle_all_none_and_the_sorted_file(tmp_path):
    cwd = make_tree(tmp_path, [(1, "so"), (0, "hor"), (1, "bm"), (0, "kmp")])
    rc, out = run(cwd)
    assert rc == 0 and "No parameter given" in out and registry(cwd).startswith("#1 #so")   # untouched (select.c:67)
    rc, out = run(cwd, "hor", "bm")                                                           # select.c:159-173
    assert "The hor algorithm has been selected" in out and "The bm algorithm has been deselected" in out
    assert registry(cwd) == "#0 #bm \n#1 #hor \n#0 #kmp \n#1 #so \n"                         # sorted by name (select.c:186-193)
    run(cwd, "-all")
    assert registry(cwd) == "#1 #bm \n#1 #hor \n#1 #kmp \n#1 #so \n"
    run(cwd, "-none", "kmp")                                                                  # left to right
    assert registry(cwd) == "#0 #bm \n#0 #hor \n#1 #kmp \n#0 #so \n"
    rc, out = run(cwd, "-which")                                                              # select.c:81-90
    assert out.split() == ["The", "list", "of", "selected", "algorithms:", "-kmp"]
    rc, out = run(cwd, "-show")                                                               # select.c:72-80
    assert out.splitlines()[1:] == ["bm", "hor", "kmp", "so"]
    before = registry(cwd)
    rc, out = run(cwd, "so", "nosuch", "bm")                                                  # select.c:185: nothing is written
    assert "no parameter nosuch" in out and registry(cwd) == before
    rc, out = run(cwd, "-h")
    assert "SMART UTILITY FOR SELECTING STRING MATCHING ALGORITHMS" in out and "-add ALGO" in out


def test_add_needs_the_executable_a_new_name_and_a_passing_test(tmp_path):
    cwd = make_tree(tmp_path, [(1, "hor"), (0, "bm")])
    rc, out = run(cwd, "-add", "epsm")
    assert "program source/bin/epsm does not exist" in out and "epsm" not in registry(cwd)   # select.c:123
    open(os.path.join(cwd, "source", "bin", "epsm"), "w").close()
    open(os.path.join(cwd, "source", "bin", "HOR"), "w").close()
    rc, out = run(cwd, "-add", "HOR")
    assert "algorithm HOR already in the set" in out                                         # select.c:100-101 (any case)
    rc, out = run(cwd, "-add", "epsm")
    assert "Testing the algorithm for correctness....ok" in out and "added succesfully" in out
    assert open(os.path.join(cwd, "tested")).read() == "epsm -nv\n"                           # select.c:106
    assert registry(cwd) == "#0 #bm \n#0 #epsm \n#1 #hor \n"                                  # added deselected (select.c:116)
    rc, out = run(cwd, "-add")
    assert "Error in input parameters. Use -h for help." in out


def test_add_refuses_an_algorithm_that_fails_its_test(tmp_path):
    cwd = make_tree(tmp_path, [(1, "hor")], test_status=1)
    open(os.path.join(cwd, "source", "bin", "kr"), "w").close()
    rc, out = run(cwd, "-add", "kr")
    assert "failed!" in out and "unable to add the algorithm kr" in out and registry(cwd) == "#1 #hor \n"   # select.c:109-112


def test_smart_reads_what_select_wrote(tmp_path):
    """The same file through the harness's reader (host/smart.c read_registry restates getAlgo): checked on the text of
    both parsers' contract — '#', flag digit, ' #', name up to the blank."""
    cwd = make_tree(tmp_path, [(0, "hor"), (0, "bm"), (0, "epsm")])
    run(cwd, "epsm", "hor")
    sel = [ln.split("#")[2].strip() for ln in registry(cwd).splitlines() if ln.startswith("#1")]
    assert sel == ["epsm", "hor"]
