#!/usr/bin/env python3
"""kernel_resources.py [FILTER ...] [-DFLAG ...] -- VGPRs, SGPR spills, scratch and LDS of every kernel of the library's units
(csrc/ptmi_*.hip; device-only -S compiles with the library's flags, in parallel); run after every kernel edit: a few bytes of scratch
in a hot loop cost tens of percent.  tests/test_kernel_resources.py holds the budgets of the render kernels."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"ptmi(_contracted)?::\(anonymous namespace\)::", "", d).split("(")[0].replace("void ", "") for d in out[:len(names)]]


def assembly(extra_flags=(), out_dir=None, units=None):
    """{unit: path of its device assembly}"""
    b = graft.load_package()._build
    out_dir = out_dir or os.path.join(ROOT, "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    extra = list(extra_flags)
    units = units or (b.KERNEL_UNITS + (b.ABLATION_UNITS if "-DPTMI_ABLATIONS" in extra else []))

    def one(u):
        asm = os.path.join(out_dir, u.replace(".hip", ".s"))
        cmd = [b.hipcc_path()] + b.COMPILE_FLAGS + extra + ["-S", "--cuda-device-only", "-o", asm, os.path.join(b.CSRC, u)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode:
            raise RuntimeError(res.stderr)
        return u, asm
    with ThreadPoolExecutor(b.JOBS) as pool:
        return dict(pool.map(one, units))


def kernels_of(asm_text):
    """[(mangled name, .amdhsa_kernel block, code of the function)]"""
    found = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm_text, re.S):
        name, body = m.group(1), m.group(2)
        fn = re.search(r"^" + re.escape(name) + r":.*?\n(.*?)\.Lfunc_end", asm_text, re.S | re.M)
        found.append((name, body, fn.group(1) if fn else ""))
    return found


def collect(extra_flags=(), out_dir=None):
    """{demangled kernel name: {unit, vgpr, scratch, lds, sgpr_spill_lanes, scratch_loads, scratch_stores}}"""
    found = {}
    for unit, asm in assembly(extra_flags, out_dir).items():
        ks = kernels_of(open(asm).read())
        for (name, body, code), demangled in zip(ks, demangle([k[0] for k in ks])):
            get = lambda k: int(re.search(k + r" (\d+)", body).group(1))   # noqa: E731
            found[demangled] = {"unit": unit, "mangled": name, "vgpr": get("next_free_vgpr"), "scratch": get("private_segment_fixed_size"),
                                "lds": get("group_segment_fixed_size"), "sgpr_spill_lanes": len(re.findall(r"v_writelane_b32", code)),
                                "scratch_loads": len(re.findall(r"scratch_load", code)), "scratch_stores": len(re.findall(r"scratch_store", code))}
    return found


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("-D")]
    for name, r in sorted(collect([a for a in sys.argv[1:] if a.startswith("-D")]).items(), key=lambda kv: (kv[1]["unit"], kv[0])):
        if want and not any(w in name or w in r["mangled"] for w in want):
            continue
        print("%-26s %-44s vgpr %3d  scratch %4d B  lds %5d B  sgpr-spill lanes %3d  scratch ld/st %d/%d" % (
            r["unit"], name[-44:], r["vgpr"], r["scratch"], r["lds"], r["sgpr_spill_lanes"], r["scratch_loads"], r["scratch_stores"]))


if __name__ == "__main__":
    main()
