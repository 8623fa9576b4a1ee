#!/usr/bin/env python3
"""isa_other.py [--json] [PHASE_STATS.json] -- where the non-arithmetic quarter of render_inline_kernel's VALU instructions sits.

rocprofv3 counts a launch's VALU wave-instructions by class (SQ_INSTS_VALU_ADD_F32, ..._MUL_F32, ...); what is in none of them -- moves,
compares, selects, lane reads, the division helpers -- is the class OTHER (profiles/rNN_valu_roofline.json: mix_wave_instr_per_launch),
24.5 % of the kernel's instructions on C2 in round 5.  This tool
  1. cuts the kernel's assembly (build/isa/ptmi_inline.s, written by tools/kernel_resources.py) into basic blocks, keeps the blocks of the
     main loop (the compiler's own loop comments), and counts every block's VALU instructions by the counters' classes, OTHER by opcode;
  2. gives every block a ROLE by its signature (the sphere test holds v_min_f32 and a ds_read_b128, its candidate path the v_sqrt_f32, the
     plane test the division, the sin/cos blocks the f64 arithmetic ...) and every role its executions per launch from the WAVE-level
     counters of a -DPTMI_PHASE_STATS / -DPTMI_SPHERE_STATS build (tools/phase_stats.py: `wave_block_executions` -- exact counts of how often
     a wave runs the block on C2, 64 spp);
  3. prints OTHER per role: static opcodes, executions, wave-instructions per launch, share -- and how the sum compares with the measured
     OTHER total of the committed PMC profile.
Blocks without a signature of their own are booked under the loop's control flow at one execution per trip: an estimate, labelled so."""
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
    if re.match(r"v_(fma|mad|fmac)_f64", o):
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


LOOPS = {}       # label -> the headers of the loops the block is in, outermost first (from the compiler's own comments)


def blocks_of(body):
    """[(label, [instruction lines])] in layout order; the entry block is labelled 'entry'."""
    out, label, cur = [], "entry", []
    lines = body.split("\n")
    for k, line in enumerate(lines):
        code = line.split(";")[0].rstrip()
        m = re.match(r"^(\.LBB\d+_\d+):", code)
        if m:
            out.append((label, cur))
            label, cur = m.group(1), []
            note = " ".join(l for l in lines[k:k + 4] if l.lstrip().startswith(";") or l is line)
            heads = re.findall(r"(?:Parent Loop|Header=|Loop Header:)\s*(BB\d+_\d+)?", note)
            own = ["BB" + label[4:]] if "Loop Header" in note else []
            LOOPS[label] = [h for h in re.findall(r"Parent Loop (BB\d+_\d+)", note)] + re.findall(r"Header=(BB\d+_\d+)", note) + own
            continue
        if not code.strip():
            continue
        if code.startswith("\t.") or code.startswith("."):
            continue
        cur.append(code.strip())
    out.append((label, cur))
    for lab, loops in LOOPS.items():                 # a block inside an inner loop names that loop's header only: the parents are the header's
        if loops:
            head = LOOPS.get(".L" + loops[0], [])
            if head and head[0] != loops[0]:
                LOOPS[lab] = head[:head.index(loops[0])] + loops if loops[0] in head else head + loops
    return out


def role_of(b, inner):
    ops = [i.split()[0] for i in b["code"]]
    has = lambda *names: any(any(o.startswith(n) for n in names) for o in ops)      # noqa: E731
    if has("v_cmp_class_f32") and has("v_sqrt_f32"):
        return "sqrt_slow"                         # the compiler's denormal-scaled square root: only when a lane holds 0 < x < 2^-96
    if has("v_min_f32") and has("v_cmp_ngt_f32") and has("v_cmp_gt_u32") and b["valu"] <= 26:
        return "sphere_test"
    if has("v_sqrt_f32"):
        return "sphere_candidate"
    if inner and has("v_cndmask_b32") and has("v_cmp_nle_f32") and b["valu"] <= 8 and not has("v_div_scale_f32"):
        return "sphere_candidate"                  # the fold update behind the square root
    if has("v_div_scale_f32") and inner:
        return "plane_test"
    if has("v_cvt_f64_f32") or has("v_cvt_i32_f64"):
        return "sincos_reduce"
    if has("v_fma_f64", "v_fmac_f64", "v_mul_f64"):
        return "sincos_polynomial"
    if has("v_div_scale_f32"):
        return "division_fallback"                 # compiler divisions behind __all() range checks (normal of a hit, sincos of inf / NaN): rare
    if has("v_rcp_f32"):
        return "hit_normal"
    return None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    stats_path = next((a for a in args if a.endswith(".json")), None)
    if stats_path is None:
        for tag in ("r06", "r05"):
            cand = os.path.join(ROOT, "profiles", "%s_phase_stats.json" % tag)
            if os.path.exists(cand) and "wave_block_executions" in open(cand).read():
                stats_path = cand
                break
    execs = None
    if stats_path:
        for line in open(stats_path):
            if "wave_block_executions" in line:
                execs = json.loads(line)["wave_block_executions"]
    measured = None
    for tag in ("r06", "r05"):
        cand = os.path.join(ROOT, "profiles", "%s_valu_roofline.json" % tag)
        if os.path.exists(cand):
            d = json.load(open(cand))
            measured = (tag, d["mix_wave_instr_per_launch"], d.get("build_id", d.get("source_hash")))
            break
    name, body = kernel_body("render_inline_kernelILb1ELi8EE")
    blocks = blocks_of(body)
    main_header = max(set(h[0] for h in LOOPS.values() if h), key=lambda hd: sum(1 for h in LOOPS.values() if h and h[0] == hd))
    table = []
    for label, code in blocks:
        loops = LOOPS.get(label, [])
        if not loops or loops[0] != main_header:
            continue
        counts = dict.fromkeys(CLASSES, 0)
        other_ops = {}
        for ins in code:
            op = ins.split()[0]
            if op.startswith("v_"):
                cls = classify(op)
                counts[cls] += 1
                if cls == "OTHER":
                    key = re.sub(r"_(e32|e64)$", "", op)
                    other_ops[key] = other_ops.get(key, 0) + 1
        table.append({"label": label, "code": code, "valu": sum(counts.values()), "counts": counts, "other_ops": other_ops, "inner": len(loops) > 1})
    roles = {}
    order = ["sphere_test", "sphere_candidate", "plane_test", "sincos_reduce", "sincos_polynomial", "hit_normal", "rest_of_loop", "sqrt_slow", "division_fallback"]
    for b in table:
        r = role_of(b, b["inner"]) or "rest_of_loop"
        roles.setdefault(r, []).append(b)
    # executions of ONE copy of a role's code per launch.  A role with k copies in the kernel (unrolled sites, the three sin/cos evaluations) runs
    # each copy (role executions / k) times; all copies hold the same instructions, so instructions per launch = mean static count x role executions
    e = execs or {}
    role_execs = {"sphere_test": e.get("sphere_tests"), "sphere_candidate": e.get("sphere_sqrt_path"), "plane_test": e.get("plane_division_path"),
                  "sincos_reduce": 3 * e["shade"] if e else None, "sincos_polynomial": 3 * e["shade"] if e else None, "hit_normal": None,
                  "rest_of_loop": e.get("trips"), "sqrt_slow": 0, "division_fallback": 0}
    what = {"sphere_test": "distanceTo @Sphere, the part every lane runs (16 f32 operations, the candidate test)", "sphere_candidate": "... its square root, t and the fold update, when a lane of the wave can be hit",
            "plane_test": "distanceTo @Plane with its IEEE division and fold update", "sincos_reduce": "sin/cos: argument reduction (f64), three per shade",
            "sincos_polynomial": "sin/cos: the two f64 polynomials, sign and swap by bit operations", "hit_normal": "hit: the sphere normal's three divisions by one length",
            "rest_of_loop": "everything else in the loop, booked at ONE execution per trip: flags and masks, frozen check and finish, restart, draws, rotation, apply_bounce",
            "sqrt_slow": "compiler's scaled square root (a lane with 0 < x < 2^-96): practically never", "division_fallback": "compiler divisions behind range checks (huge or zero operands, inf / NaN): practically never"}
    # copies per role: sphere and plane tests are counted per TEST, a test runs one copy; sin/cos roles per evaluation
    out_rows, total = [], 0.0
    for r in order:
        bs = roles.get(r, [])
        if not bs:
            continue
        valu = sum(b["valu"] for b in bs)
        other = sum(b["counts"]["OTHER"] for b in bs)
        ops = {}
        for b in bs:
            for k, v in b["other_ops"].items():
                ops[k] = ops.get(k, 0) + v
        if r == "sphere_test":
            copies = len(bs)
        elif r == "sphere_candidate":
            copies = max(1, len(roles.get("sphere_test", [])))
        elif r == "plane_test":
            copies = max(1, sum(1 for b in bs if any(i.startswith("v_div_fixup") for i in b["code"])))
        elif r in ("sincos_reduce", "sincos_polynomial"):
            copies = 3
        else:
            copies = 1
        if r == "hit_normal":
            n_exec = e.get("trace") if e else None               # at most once per trace (a sphere hit with a normal to normalise)
        else:
            n_exec = role_execs.get(r)
        dyn = None if n_exec is None else other / copies * n_exec
        dyn_valu = None if n_exec is None else valu / copies * n_exec
        if dyn:
            total += dyn
        out_rows.append({"role": r, "what": what[r], "blocks": len(bs), "copies": copies, "static_valu": valu, "static_other": other, "other_opcodes": ops,
                         "executions_per_launch": n_exec, "other_wave_instr_per_launch": dyn, "valu_wave_instr_per_launch": dyn_valu})
    print("kernel ...%s: main loop %s, %d blocks; wave-level executions from %s" % (name[-48:], main_header, len(table), stats_path))
    print("%-18s %4s %6s %6s %14s %14s %7s  %s" % ("role", "blk", "VALU", "OTHER", "executions", "OTHER / launch", "share", "OTHER opcodes (static, all copies)"))
    for row in out_rows:
        dyn = row["other_wave_instr_per_launch"]
        print("%-18s %4d %6d %6d %14s %14s %6s  %s" % (row["role"], row["blocks"], row["static_valu"], row["static_other"],
              "%.0f" % row["executions_per_launch"] if row["executions_per_launch"] is not None else "-", "%.0f" % dyn if dyn is not None else "-",
              "%.1f%%" % (100.0 * dyn / total) if dyn else "-", " ".join("%s:%d" % kv for kv in sorted(row["other_opcodes"].items(), key=lambda kv: -kv[1]))))
    if measured:
        print("sum %.0f M OTHER wave-instructions per launch; the PMC profile profiles/%s_valu_roofline.json (binary %s) measured %.0f M" % (total / 1e6, measured[0], measured[2], measured[1]["OTHER"] / 1e6))
    if "--json" in sys.argv:
        print(json.dumps({"kernel": name, "main_loop": main_header, "wave_block_executions": execs, "roles": out_rows, "sum_other": total,
                          "measured_other": measured[1]["OTHER"] if measured else None, "measured_in": measured[0] if measured else None}))


if __name__ == "__main__":
    main()
