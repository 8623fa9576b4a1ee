#!/usr/bin/env python3
"""kernel_resources.py [FILTER ...] [-DFLAG ...] -- VGPRs, SGPR spills, scratch and LDS of every kernel in ptmi_kernels.hip
(device-only -S compile with the library's flags); run after every kernel edit: a few bytes of scratch in a hot loop cost tens
of percent.  tests/test_kernel_resources.py holds the budgets of the render kernels."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def collect(extra_flags=(), out_dir=None):
    """{demangled kernel name: {vgpr, scratch, lds, sgpr_spill_lanes, scratch_loads, scratch_stores}}"""
    b = graft.load_package()._build
    out_dir = out_dir or os.path.join(ROOT, "build", "isa")
    os.makedirs(out_dir, exist_ok=True)
    asm = os.path.join(out_dir, "kernels.s")
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC", "-pthread", "-ldl")]
    cmd = [b.hipcc_path()] + flags + list(extra_flags) + ["-S", "--cuda-device-only", "-o", asm, os.path.join(b.CSRC, "ptmi_kernels.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        raise RuntimeError(res.stderr)
    text = open(asm).read()
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        get = lambda k: int(re.search(k + r" (\d+)", body).group(1))   # noqa: E731
        fn = re.search(r"^" + re.escape(name) + r":.*?\n(.*?)\.Lfunc_end", text, re.S | re.M)
        code = fn.group(1) if fn else ""
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        demangled = re.sub(r"ptmi::\(anonymous namespace\)::", "", demangled).split("(")[0].replace("void ", "")
        found[demangled] = {"mangled": name, "vgpr": get("next_free_vgpr"), "scratch": get("private_segment_fixed_size"),
                            "lds": get("group_segment_fixed_size"), "sgpr_spill_lanes": len(re.findall(r"v_writelane_b32", code)),
                            "scratch_loads": len(re.findall(r"scratch_load", code)), "scratch_stores": len(re.findall(r"scratch_store", code))}
    return found


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("-D")]
    for name, r in collect([a for a in sys.argv[1:] if a.startswith("-D")]).items():
        if want and not any(w in name or w in r["mangled"] for w in want):
            continue
        print("%-48s vgpr %3d  scratch %4d B  lds %5d B  sgpr-spill lanes %3d  scratch ld/st %d/%d" % (
            name[-48:], r["vgpr"], r["scratch"], r["lds"], r["sgpr_spill_lanes"], r["scratch_loads"], r["scratch_stores"]))


if __name__ == "__main__":
    main()
