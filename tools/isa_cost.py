#!/usr/bin/env python3
"""isa_cost.py -- prices a gfx950 kernel's instruction stream with the issue costs measured by
tools/valu_rates.hip on MI355X (cycles per wave-instruction per SIMD at >= 4 waves/SIMD):
  2: v_add/sub/mul_f32, v_mov, v_xor/and/or, v_add_u32, shifts      4: v_fma/fmac, v_pk_*, f64 mul/add/fma,
  cvt, v_cmp*, v_cndmask, v_max/min, v_lshl_add, v_alignbit, v_div_*  8: v_sqrt/v_rcp/v_rsq
usage: isa_cost.py file.s kernel_symbol_substring   -> per basic block: VALU instr, est. cycles"""
import re
import sys

FAST = re.compile(r"^v_(add|sub|subrev|mul)_f32|^v_mov_b32|^v_(xor|and|or|not)_b32|^v_(add|sub|subrev)_u32|^v_(lshrrev|lshlrev|ashrrev)_b32|^v_mul_legacy")
SLOW8 = re.compile(r"^v_(sqrt|rcp|rsq|exp|log|sin|cos)_f32|^v_rcp_iflag")
F64_8 = re.compile(r"^v_(rcp|sqrt|rsq|div_scale|div_fmas|div_fixup)_f64")


def cost(op):
    if not op.startswith("v_"):
        return 0
    if FAST.match(op):
        return 2
    if SLOW8.match(op) or F64_8.match(op):
        return 8
    if op.startswith("v_mul_lo_u32") or op.startswith("v_mul_hi_u32"):
        return 8
    return 4


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and ":" in l)
    block, blocks, order = "entry", {}, []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            block = m.group(1)
            continue
        m = re.match(r"^([a-z][a-z0-9_]+)", t)
        if not m:
            continue
        op = m.group(1)
        if block not in blocks:
            blocks[block] = [0, 0, 0, 0]
            order.append(block)
        b = blocks[block]
        if op.startswith("v_"):
            b[0] += 1
            b[1] += cost(op)
        elif op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"):
            b[2] += 1
        else:
            b[3] += 1
    tot = [0, 0, 0, 0]
    for k in order:
        b = blocks[k]
        print("%-12s valu %4d  est.cycles %5d  mem %3d  salu/other %4d" % (k, *b))
        tot = [x + y for x, y in zip(tot, b)]
    print("%-12s valu %4d  est.cycles %5d  mem %3d  salu/other %4d" % ("TOTAL", *tot))


if __name__ == "__main__":
    main()
