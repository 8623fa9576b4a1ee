#!/usr/bin/env python3
"""isa_diff.py BASELINE.s [-DFLAG ...] -- is the code of every kernel of the current units the code it was in BASELINE.s (a device
assembly made earlier, e.g. of the sources before a refactoring)?  Kernels are matched by demangled name (template arguments
normalised by NAME_MAP below); instructions are compared after local labels and symbol hashes are normalised.  Used in round 4 to
show that splitting ptmi_kernels.hip into one unit per kernel family, and moving the diagnostics into probes, changed no kernel."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

NAME_MAP = [(r"render_inline_kernel<(true|false), 0, (\d+)>", r"render_inline_kernel<\1, \2>"),          # round 3 -> round 4 names
            (r"render_inline_kernel<(true|false), 0>", r"render_inline_kernel<\1, 0>"),
            (r"render_inline_kernel<(true|false), ([123]), (\d+)>", r"render_inline_modes_kernel<\1, \2, \3>")]


def normalise(code):
    out = []
    for line in code.split("\n"):
        line = line.split(";")[0].strip()
        if not line or line.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", line):
                out.append("L:")
            continue
        line = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", line)
        line = re.sub(r"_ZN\S+", "SYM", line)
        line = re.sub(r"__unnamed_\d+|\.str\.\d+", "STR", line)
        out.append(line)
    return out


def kernels(path_or_text, is_path=True):
    text = open(path_or_text).read() if is_path else path_or_text
    ks = kr.kernels_of(text)
    res = {}
    for (name, body, code), dem in zip(ks, kr.demangle([k[0] for k in ks])):
        for pat, rep in NAME_MAP:
            dem = re.sub(pat, rep, dem)
        res[dem] = normalise(code)
    return res


def main():
    base = kernels(sys.argv[1])
    flags = [a for a in sys.argv[2:] if a.startswith("-")]
    units = None
    if "-DPTMI_CONTRACTED_BUILD" in flags:
        units = ["ptmi_inline.hip"]
    b = kr.graft.load_package()._build
    if "-DPTMI_CONTRACTED_BUILD" in flags:     # the contracted object's flags
        saved = b.COMPILE_FLAGS
        b.COMPILE_FLAGS = [f for f in saved if f != "-ffp-contract=off"]
    now = {}
    for unit, asm in kr.assembly(flags, os.path.join(ROOT, "build", "isa_now"), units).items():
        now.update(kernels(asm))
    same, differ, missing = 0, [], []
    for name, code in sorted(base.items()):
        if name not in now:
            missing.append(name)
        elif now[name] == code:
            same += 1
        else:
            a, c = code, now[name]
            first = next((i for i in range(min(len(a), len(c))) if a[i] != c[i]), min(len(a), len(c)))
            differ.append((name, len(a), len(c), first, a[first] if first < len(a) else None, c[first] if first < len(c) else None))
    print("%d kernels identical, %d differ, %d missing, %d new" % (same, len(differ), len(missing), len(set(now) - set(base))))
    for d in differ:
        print("  DIFFERS %s: %d vs %d lines, first at %d: %r | %r" % d)
    for m in missing:
        print("  MISSING", m)
    for n in sorted(set(now) - set(base)):
        print("  NEW    ", n)
    return 1 if differ or missing else 0


if __name__ == "__main__":
    sys.exit(main())
