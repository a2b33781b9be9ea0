"""Which source files decide the code of which kernel (smart_amd/csrc: one translation unit — one code object —
per kernel family), and a sha256 over such a set.  Measurements that are only valid for the kernel source they were
taken on (profiles/pmc_traffic.json: roofline.traffic of bench.py) are bound to the FAMILY's sha, so an edit to
bm_scan does not void the PMC traffic of so_runs."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# every kernel unit includes these (device helpers, argument structs, launch geometry)
COMMON = ("dev_common.hpp", "kernels.hpp", "launch_common.hpp")
UNITS = {
    "k_hor": ("k_hor.hip",),
    "k_horg": ("k_horg.hip", "gram_skip.hpp"),
    "k_bmg": ("k_bmg.hip", "gram_skip.hpp"),
    "k_bm": ("k_bm.hip",),
    "k_bndm": ("k_bndm.hip",),
    "k_bndmx": ("k_bndmx.hip",),
    "k_so": ("k_so.hip", "runs_common.hpp"),
    "k_kmp": ("k_kmp.hip", "runs_common.hpp"),
    "k_packed": ("k_packed.hip",),
    "k_util": ("k_util.hip",),
}
KERNEL_UNIT = {
    "hor_scan": "k_hor", "hor_scan_bp": "k_hor", "hor_scan_gram": "k_horg", "bm_scan_gram": "k_bmg", "bm_scan": "k_bm", "bndm_scan": "k_bndm", "sbndm_scan": "k_bndmx",
    "bndml_scan": "k_bndmx", "so_runs": "k_so", "kmp_runs": "k_kmp", "packed_scan": "k_packed", "packed_find": "k_packed",
    "generate_text": "k_util", "tile_fill": "k_util", "text_alphabet": "k_util", "probe_read": "k_util",
}


def unit_files(unit):
    return [os.path.join(CSRC, f) for f in UNITS[unit] + COMMON]


def unit_sha256(unit, read=None):
    """sha256 over the unit's own sources and the common headers, in a fixed order.  `read(path) -> bytes`
    replaces the file system (tools/collect_profiles.py hashes the COMMITTED sources through git show)."""
    h = hashlib.sha256()
    for p in unit_files(unit):
        data = read(p) if read else open(p, "rb").read()
        h.update(os.path.basename(p).encode() + b"\0" + data + b"\0")
    return h.hexdigest()


def kernel_sha256(kernel, read=None):
    return unit_sha256(KERNEL_UNIT[kernel], read)


def all_unit_shas(read=None):
    return {u: unit_sha256(u, read) for u in UNITS}
