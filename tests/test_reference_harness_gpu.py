"""Process-level drop-in check: the REFERENCE's own harness binaries (src/test.c and
src/smart.c compiled as they are into oracle/_ref/bin, where /root/reference exists)
drive this project's plugin executables through SMART's unchanged shm/exec protocol
(src/smart.c:140-146, src/algos/include/main.h:42-122)."""
import os
import re
import shutil
import subprocess
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REFBIN = os.path.join(ROOT, "oracle", "_ref", "bin")
PLUGINS = os.path.join(ROOT, "smart_amd", "bin", "plugins")
def run_group(cmd, cwd, env, timeout):
    """Run `cmd` in its own process group and end the whole group on a timeout (the reference spawns the
    plugins through system(): killing only the harness would leave a plugin behind, holding the GPU).
    Returns (stdout, stderr, timed_out)."""
    import signal
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return out, err, False
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        return out, err + "\n[timed out after %d s]" % timeout, True


ALGOS = ["hor", "bm", "kmp", "so", "bndm", "epsm", "sa", "qs", "tunedbm", "raita", "hash3", "hash5", "hash8", "sbndm", "kr", "bndml"]


@pytest.fixture(scope="module")
def smart_tree(tmp_path_factory, oracle):
    if not os.path.exists(os.path.join(REFBIN, "test")):
        pytest.skip("oracle/_ref/bin not built (no /root/reference on the build host)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "smart_amd", "host")])
    # some kernels refuse SysV shm; then this check cannot run here
    probe = subprocess.run(["ipcs", "-m"], capture_output=True)
    if probe.returncode != 0:
        pytest.skip("SysV shared memory not available")
    d = tmp_path_factory.mktemp("smart_tree")
    # the layout the reference hard-codes (SURVEY.md §2 "Layout drift")
    os.makedirs(d / "source" / "bin")
    for a in ALGOS:
        shutil.copy(os.path.join(PLUGINS, a), d / "source" / "bin" / a)
    # executables copied out of the tree: point their rpath-relative library lookup at the real one
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "smart_amd", "csrc") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    with open(d / "source" / "algorithms.h", "w") as f:
        for a in sorted(ALGOS):
            f.write("#1 #%s \n" % a)
    os.makedirs(d / "data" / "rand128")
    with open(d / "data" / "rand128" / "index.txt", "w") as f:
        f.write("random text over 128 chars\n#rand128.txt#\n")
    oracle.textgen(128, 2 * 1048576).tofile(str(d / "data" / "rand128" / "rand128.txt"))
    os.makedirs(d / "results")
    for tool in ("test", "smart"):
        shutil.copy(os.path.join(REFBIN, tool), d / tool)
    return d, env


@pytest.mark.parametrize("algo", ALGOS)
def test_reference_test_binary_passes(smart_tree, algo):
    d, env = smart_tree
    # The reference draws its five SysV keys from rand() % 1000 seeded with time(NULL) and the loop that
    # draws the COUNT segment's key (src/test.c:205-209) compares pkey, not rkey, with ekey/prekey: about
    # 2 runs in 1000 the count and a timing segment are the same memory and the reference reads a garbage
    # count (seen once: "found 0 occ instead of 10" on case 1).  A failed run is repeated after the seed
    # has changed; a real defect fails every time.
    # (Seen once as well: a run that never returned.  A run gets 60 s — it takes 6 — and counts as failed.)
    for attempt in range(3):
        out, err, _ = run_group(["./test", algo], str(d), env, 60)
        if "Well done! Test passed successfully" in out:
            break
        time.sleep(1.1)
    assert "Well done! Test passed successfully" in out, out + err


def test_reference_smart_binary_reports_ok(smart_tree):
    d, env = smart_tree

    def ok(out):
        lines = [ln for ln in out.splitlines() if re.search(r"\] [A-Z0-9]+ \.", ln)]
        return len(lines) == len(ALGOS) and all("[OK]" in ln for ln in lines)

    for attempt in range(3):  # same key-collision hazard as above (src/smart.c:262-267)
        out, err, _ = run_group(["./smart", "-text", "rand128", "-plen", "32", "32", "-pset", "3", "-occ", "-pre"],
                                str(d), env, 150)
        if ok(out):
            break
        time.sleep(1.1)
    assert "Testing %d algorithms" % len(ALGOS) in out, out + err
    for a in ALGOS:
        line = [ln for ln in out.splitlines() if re.search(r"\] %s \." % a.upper(), ln)]
        assert line and "[OK]" in line[0] and re.search(r"occ [1-9]", line[0]), (a, out)
