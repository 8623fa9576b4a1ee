#!/usr/bin/env python3
"""kernel_resources.py [FILTER] -- VGPRs, SGPR spills, scratch and LDS of every kernel in ptmi_kernels.hip (device-only -S
compile with the library's flags); run after every kernel edit: a few bytes of scratch in a hot loop cost tens of percent."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
b = pkg._build
out = os.path.join(ROOT, "build", "isa")
os.makedirs(out, exist_ok=True)
flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC", "-pthread", "-ldl")]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
cmd = [b.hipcc_path()] + flags + extra + ["-S", "--cuda-device-only", "-o", os.path.join(out, "kernels.s"), os.path.join(b.CSRC, "ptmi_kernels.hip")]
res = subprocess.run(cmd, capture_output=True, text=True)
if res.returncode:
    sys.exit(res.stderr)
text = open(os.path.join(out, "kernels.s")).read()
want = [a for a in sys.argv[1:] if not a.startswith("-D")]
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name, body = m.group(1), m.group(2)
    if want and not any(w in name for w in want):
        continue
    get = lambda k: re.search(k + r" (\d+)", body).group(1)
    fn = re.search(r"^" + re.escape(name) + r":.*?\n(.*?)\.Lfunc_end", text, re.S | re.M)
    code = fn.group(1) if fn else ""
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    demangled = re.sub(r"ptmi::\(anonymous namespace\)::", "", demangled).split("(")[0]
    print("%-52s vgpr %3s  scratch %4s B  lds %5s B  sgpr-spill lanes %3d  scratch ld/st %d/%d" % (
        demangled[-52:], get("next_free_vgpr"), get("private_segment_fixed_size"), get("group_segment_fixed_size"),
        len(re.findall(r"v_writelane_b32", code)), len(re.findall(r"scratch_load", code)), len(re.findall(r"scratch_store", code))))
