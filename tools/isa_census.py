#!/usr/bin/env python3
"""Instruction census of a kernel's hot basic block, from the ISA hipcc emits (development tool; runs without a GPU).

    python tools/isa_census.py depthmodelhardening_amd/csrc/wino_conv.hip 'wino_conv_kernelILi32ELb0ELb1ELb0E'

Compiles the file for gfx950 with the library's flags (device side only, -S), takes the kernel whose mangled name contains the
given substring and prints, for each basic block with at least 32 MFMAs, the number of instructions per class -- the
steady-state chunk of K10 is such a block (plus the four short write / load slots that follow it, counted with --tail N lines).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def klass(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_read", "ds_load")):
        return op.split("_")[0] + "_" + "_".join(op.split("_")[1:])
    if op.startswith(("ds_write", "ds_store")):
        return op
    if op.startswith("buffer_load"):
        return "buffer_load ... lds" if ins.rstrip().endswith(" lds") else "buffer_load"
    if op.startswith(("buffer_store", "global_store")):
        return "store"
    if op.startswith("global_load"):
        return "global_load"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("v_accvgpr"):
        return "v_accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op in ("s_waitcnt", "s_barrier", "s_nop"):
        return op
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
                        "-I" + os.path.join(REPO, "include"), "-I" + os.path.dirname(os.path.abspath(src)),
                        "--offload-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    print(lines[start].rstrip(":"))
    name, cur, blocks = "entry", [], []
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
        else:
            t = l.strip()
            if t and not t.startswith((";", ".")):
                cur.append(t)
    blocks.append((name, cur))
    for name, ins in blocks:
        c = collections.Counter(klass(i) for i in ins)
        if c["mfma"] >= 32:
            print("%s: %d instructions" % (name, len(ins)))
            for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
                print("    %-22s %4d" % (k, v))


if __name__ == "__main__":
    main()
