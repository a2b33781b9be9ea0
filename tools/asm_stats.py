#!/usr/bin/env python3
"""Per-kernel facts from hipcc's gfx950 assembly (-save-temps=obj: *-hip-amdgcn-amd-amdhsa-gfx950.s):
instructions, VGPRs, SGPRs, scratch bytes, and a hash of the instruction stream (labels and comments stripped),
so that two builds can be compared kernel by kernel — "did moving this kernel to its own translation unit change
its code?" — without a GPU.

    python tools/asm_stats.py a.s [b.s ...]            one table per file
    python tools/asm_stats.py --diff old.s new1.s new2.s ...   kernels of the new files against the old one
"""
import hashlib
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def parse(path):
    kernels = {}
    cur = None
    body = []
    meta = {}
    with open(path) as f:
        for line in f:
            s = line.strip()
            m = re.match(r"^\.amdhsa_kernel\s+(\w+)", s)
            if m:
                meta[m.group(1)] = {}
                continue
            m = re.match(r"^\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|accum_offset)\s+(\d+)", s)
            if m and meta:
                meta[list(meta)[-1]][m.group(1)] = int(m.group(2))
                continue
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", s)
            if m and cur is None:
                cur = m.group(1)
                body = []
                continue
            if cur is not None:
                if s.startswith(".Lfunc_end"):
                    kernels[cur] = {"body": body}
                    cur = None
                    continue
                if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                    continue
                body.append(re.sub(r"\s*;.*$", "", s))
    names = demangle(list(kernels))
    out = {}
    for k, v in kernels.items():
        if k not in meta:
            continue  # a device function, not a kernel
        code = "\n".join(re.sub(r"\.LBB\d+_\d+", "L", x) for x in v["body"])
        name = re.sub(r"^void sg::", "", names[k])
        name = re.sub(r"\(sg::ScanArgs.*$", "", name)
        out[name] = {"insts": len(v["body"]), "vgpr": meta[k].get("next_free_vgpr"), "sgpr": meta[k].get("next_free_sgpr"),
                     "scratch": meta[k].get("private_segment_fixed_size"), "hash": hashlib.sha1(code.encode()).hexdigest()[:10]}
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == "--diff":
        old = parse(args[1])
        new = {}
        for p in args[2:]:
            new.update(parse(p))
        print("%-58s %7s %7s %5s %5s %4s %4s  %s" % ("kernel", "insts", "new", "vgpr", "new", "scr", "new", "code"))
        for k in sorted(set(old) | set(new)):
            o, n = old.get(k), new.get(k)
            if o is None or n is None:
                print("%-58s %s" % (k[:58], "only in the NEW build" if o is None else "only in the OLD build"))
                continue
            same = "same" if o["hash"] == n["hash"] else "differs"
            print("%-58s %7d %7d %5s %5s %4s %4s  %s" % (k[:58], o["insts"], n["insts"], o["vgpr"], n["vgpr"], o["scratch"], n["scratch"], same))
        return
    for p in args:
        print("==", p)
        for k, v in sorted(parse(p).items()):
            print("%-58s insts %6d vgpr %4s sgpr %4s scratch %4s  %s" % (k[:58], v["insts"], v["vgpr"], v["sgpr"], v["scratch"], v["hash"]))


if __name__ == "__main__":
    main()
