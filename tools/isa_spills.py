#!/usr/bin/env python3
"""isa_spills.py KERNEL_SUBSTRING [N] -- after tools/kernel_resources.py has written build/isa/*.s (one per unit): where the N-th
kernel whose mangled name contains the substring touches scratch memory (line within the kernel, basic block, instruction)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def body_of(sub, nth=0):
    import glob
    t = "\n".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "build", "isa", "*.s")))).split("\n")
    starts = [i for i, l in enumerate(t) if l.startswith("_ZN") and sub in l and l.split(":")[0].endswith("E") and ":" in l and not l.startswith("\t")]
    start = starts[nth]
    end = next(i for i in range(start, len(t)) if "s_endpgm" in t[i])
    return t[start].split(":")[0], t[start:end]


if __name__ == "__main__":
    name, body = body_of(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(name, len(body), "lines")
    lab = None
    for i, l in enumerate(body):
        if re.match(r"^\.LBB\d+_\d+:", l):
            lab = l.split(":")[0]
        if "scratch_" in l:
            print(i, lab, l.strip())
