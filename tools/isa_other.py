#!/usr/bin/env python3
"""isa_other.py [KERNEL_SUBSTRING] [PROFILE.json] -- where the non-arithmetic quarter of a kernel's VALU instructions sits.

rocprofv3 counts a launch's VALU wave-instructions by class (SQ_INSTS_VALU_ADD_F32, ..._MUL_F32, ...); what is in none of them -- moves,
compares, selects, lane reads, bit-field ops -- is the class OTHER (profiles/rNN_valu_roofline.json: mix_wave_instr_per_launch), 24.5 % of
render_inline_kernel's instructions on C2.  This tool
  1. cuts the kernel's assembly (build/isa/ptmi_inline.s, written by tools/kernel_resources.py) into basic blocks and counts every block's
     VALU instructions by the same classes, OTHER by opcode;
  2. estimates how often a wave executes each block per launch from the MEASURED class totals: non-negative least squares of
     (class count of block b) x (executions of b) = (class total of the launch) over the blocks of the main loop -- twelve equations, the
     blocks have distinct signatures (the sphere test holds the sqrt, the plane test the division, the shade the f64 sin/cos ...);
  3. prints OTHER per block: static opcodes, estimated executions, share of the launch's OTHER instructions.
The estimate is a fit, not a trace: it says which blocks to look at, and how well the fit reproduces the class totals is printed with it."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CLASSES = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "CVT", "INT32", "INT64", "OTHER"]


def classify(op):
    """The counter class of one VALU opcode (gfx950 SQ_INSTS_VALU_* semantics as far as the guide states them; what matches none is OTHER)."""
    o = op
    for suffix in ("_e32", "_e64", "_dpp", "_sdwa"):
        if o.endswith(suffix):
            o = o[:-len(suffix)]
    if re.match(r"v_(pk_)?(fma|mad|mac|fmac|fmaak|fmamk|dot\d)_(f32|legacy_f32)", o) or o in ("v_fma_mix_f32",):
        return "FMA_F32"
    if re.match(r"v_(pk_)?(add|sub|subrev|min|max|min3|max3|med3)_f32", o):
        return "ADD_F32"
    if re.match(r"v_(pk_)?(mul|mul_legacy)_f32", o):
        return "MUL_F32"
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log|rcp_iflag)_(f32|legacy_f32)", o):
        return "TRANS_F32"
    if re.match(r"v_(fma|mad)_f64", o):
        return "FMA_F64"
    if re.match(r"v_(add|min|max)_f64", o):
        return "ADD_F64"
    if re.match(r"v_mul_f64", o):
        return "MUL_F64"
    if re.match(r"v_(rcp|rsq|sqrt)_f64", o):
        return "TRANS_F64"
    if re.match(r"v_cvt_", o):
        return "CVT"
    if re.match(r"v_(add|sub|subrev|mul|mad|lshl|lshr|ashr)\w*_(u64|i64|b64)", o) or o in ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_mad_u64_u32", "v_mad_i64_i32"):
        return "INT64"
    if re.match(r"v_(add|sub|subrev|addc|subb|mul_lo|mul_hi|mul|mad|lshl|lshlrev|lshrrev|ashrrev|and|or|xor|xnor|not|bfe|bfi|alignbit|alignbyte|min|max|min3|max3|med3|"
                r"add3|lshl_add|add_lshl|lshl_or|and_or|or3|xad|perm|bcnt|ffbh|ffbl|sad)\w*_(u32|i32|b32|u16|i16|b16|u24|i24|co_u32|co_ci_u32)", o):
        return "INT32"
    return "OTHER"


def kernel_body(sub):
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "build", "isa", "*.s"))):
        text = open(path).read()
        for m in re.finditer(r"^(_Z\S*):\s*;?.*?\n(.*?)\.Lfunc_end", text, re.S | re.M):
            name = m.group(1)
            if sub in name and "render" in name:
                return name, m.group(2)
    raise SystemExit("no kernel matching %r in build/isa/*.s (run tools/kernel_resources.py first)" % sub)


def blocks_of(body):
    """[(label, [instruction lines])] in layout order; the entry block is labelled 'entry'."""
    out, label, cur = [], "entry", []
    for line in body.split("\n"):
        code = line.split(";")[0].rstrip()
        if not code.strip():
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", code)
        if m:
            out.append((label, cur))
            label, cur = m.group(1), []
            continue
        if code.startswith("\t.") or code.startswith("."):
            continue
        cur.append(code.strip())
    out.append((label, cur))
    return out


def main():
    sub = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].endswith(".json") else "render_inline_kernelILb1ELi8EE"
    profile = next((a for a in sys.argv[1:] if a.endswith(".json")), None)
    if profile is None:
        for tag in ("r06", "r05"):
            cand = os.path.join(ROOT, "profiles", "%s_valu_roofline.json" % tag)
            if os.path.exists(cand):
                profile = cand
                break
    name, body = kernel_body(sub)
    blocks = blocks_of(body)
    table = []
    for label, code in blocks:
        counts = dict.fromkeys(CLASSES, 0)
        other_ops = {}
        marks = set()
        for ins in code:
            op = ins.split()[0]
            if op.startswith("v_"):
                cls = classify(op)
                counts[cls] += 1
                if cls == "OTHER":
                    key = re.sub(r"_(e32|e64)$", "", op)
                    other_ops[key] = other_ops.get(key, 0) + 1
            if op.startswith("ds_"):
                marks.add("lds")
            if op.startswith("global_") or op.startswith("buffer_"):
                marks.add("global")
            if op.startswith("scratch_"):
                marks.add("scratch")
        table.append({"label": label, "n": len(code), "valu": sum(counts.values()), "counts": counts, "other_ops": other_ops, "marks": sorted(marks),
                      "branches": [i.split()[-1] for i in code if i.startswith("s_cbranch") or i.startswith("s_branch")]})
    mix = json.load(open(profile))["mix_wave_instr_per_launch"] if profile else None
    est = None
    if mix:
        from scipy.optimize import nnls
        hot = [i for i, b in enumerate(table) if b["valu"] > 0]
        A = np.array([[table[i]["counts"][c] for i in hot] for c in CLASSES], float)
        y = np.array([mix.get(c, 0.0) for c in CLASSES], float)
        scale = 1.0 / np.maximum(y, y.max() * 1e-3)               # relative residuals: the small classes (sqrt, f64) carry the signatures
        x, _ = nnls(A * scale[:, None], y * scale)
        est = dict(zip(hot, x))
        fit = A @ x
        print("fit of the class totals (launch, wave-instructions; fitted / measured):")
        print("  " + "  ".join("%s %.3f" % (c, (f / m) if m else float("nan")) for c, f, m in zip(CLASSES, fit, y)))
    total_other = sum((est or {}).get(i, 0.0) * b["counts"]["OTHER"] for i, b in enumerate(table)) or 1.0
    print("kernel %s: %d blocks, %d VALU instructions static, %d of them OTHER" % (name[-60:], len(table), sum(b["valu"] for b in table), sum(b["counts"]["OTHER"] for b in table)))
    print("%-12s %5s %5s %6s %14s %7s  %s" % ("block", "instr", "VALU", "OTHER", "est. execs", "% OTHER", "OTHER opcodes | marks"))
    rows = sorted(range(len(table)), key=lambda i: -((est or {}).get(i, 0.0) * table[i]["counts"]["OTHER"]))
    for i in rows[:24]:
        b = table[i]
        e = (est or {}).get(i, 0.0)
        print("%-12s %5d %5d %6d %14.0f %6.1f%%  %s | %s" % (b["label"], b["n"], b["valu"], b["counts"]["OTHER"], e, 100.0 * e * b["counts"]["OTHER"] / total_other,
                                                          " ".join("%s:%d" % kv for kv in sorted(b["other_ops"].items(), key=lambda kv: -kv[1])), ",".join(b["marks"])))
    if "--json" in sys.argv:
        print(json.dumps({"kernel": name, "blocks": [{k: b[k] for k in ("label", "n", "valu", "counts", "other_ops", "marks")} | {"est_execs": (est or {}).get(i)} for i, b in enumerate(table)]}))


if __name__ == "__main__":
    main()
