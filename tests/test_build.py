"""Properties of the compiled gfx950 code objects that the measurements depend on (no GPU needed: hipcc cross-compiles)."""
import os
import re
import subprocess

from conftest import ROOT

CSRC = os.path.join(ROOT, "smart_amd", "csrc")


from smart_amd import sources  # noqa: E402

KERNEL_UNITS = sorted(sources.UNITS)  # one translation unit per kernel family


def hipcc_units(extra, out_of):
    """Run hipcc over every kernel unit (in parallel) with `extra` flags; out_of(unit) -> output path or None."""
    procs = []
    for u in KERNEL_UNITS:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only"] + extra + \
              ["-o", out_of(u) or "/dev/null", os.path.join(CSRC, u + ".hip")]
        procs.append((u, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    res = {}
    for u, p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, (u, se[-2000:])
        res[u] = se
    return res


def resource_usage():
    """{kernel: {"vgprs": n, "scratch": bytes per lane, "lds": static bytes per workgroup}} from -Rpass-analysis=kernel-resource-usage."""
    stderr = "\n".join(hipcc_units(["-Rpass-analysis=kernel-resource-usage", "-c"], lambda u: None).values())
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
        m = re.search(r"\bVGPRs: (\d+)", line)
        if m and cur:
            out[cur]["vgprs"] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            out[cur]["scratch"] = int(m.group(1))
        m = re.search(r"LDS Size \[bytes/block\]: (\d+)", line)
        if m and cur:
            out[cur]["lds"] = int(m.group(1))
    return out


def test_no_product_kernel_uses_scratch():
    """A kernel that uses scratch at all — a few spilled registers far from the hot loop — measured 12 % slower on
    1 GiB (so_runs<LONG>): every kernel of the product library must compile without it, and the
    1024-thread runs kernels within the 128 VGPRs their 16 waves per CU allow."""
    usage = resource_usage()
    scan = {k: v for k, v in usage.items() if re.search(r"(_scan|_runs|_find|_gram)I", k)}
    assert len(scan) >= 30, sorted(usage)
    for k, v in scan.items():
        assert v.get("scratch") == 0, (k, v)
    runs = {k: v for k, v in scan.items() if re.search(r"(so_runs|kmp_runs)I", k)}
    assert len(runs) == 8, sorted(runs)  # so_runs<LONG, FOUR> x 4, kmp_runs<PREFIX, FOUR> x 4
    for k, v in runs.items():
        assert v["vgprs"] <= 128, (k, v)
    # No STATIC LDS in any scan kernel: hor_flat, bm_scan, bndm_scan and kmp_runs address LDS by absolute offset (the
    # dynamic segment must start at offset 0; a static __shared__ in a helper would move it, the kernels would poison
    # their count and the library would report an error — api.cpp count_poisoned — instead of counting).
    for k, v in scan.items():
        assert v.get("lds") == 0, (k, v)


def test_kernels_load_the_text_with_global_instructions(tmp_path):
    """Text and table pointers must stay recognisable as global memory: a kernel whose pointers lose that
    (round 2: a select between two argument structs) loads with flat_load and computes addresses on the vector
    unit — packed_scan ran 12-15 % slower.  A handful of flat loads remain in rarely taken verification code."""
    hipcc_units(["-S"], lambda u: str(tmp_path / (u + ".s")))
    text = "\n".join((tmp_path / (u + ".s")).read_text() for u in KERNEL_UNITS)
    flat, glob = len(re.findall(r"\bflat_load", text)), len(re.findall(r"\bglobal_load", text))
    assert glob > 300 and flat < 0.1 * glob, (flat, glob)
