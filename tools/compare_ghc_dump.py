#!/usr/bin/env python3
"""compare_ghc_dump.py DUMP [--json OUT] [--no-device] -- the MI355X side of the parity experiment.

DUMP is what `ptmi-dump` (haskell-path-tracer_amd/haskell/dump/Dump.hs, run by a maintainer who has a GHC build of the reference)
wrote: the seed words it injected, the generator states `createWith` made of them, raw word and float draws from 64 of those
states, and the seven planes after one and two calls of `render Inline` and of `render Streams` on Accelerate's CPU backend.
This tool renders the same inputs through the oracle (oracle/pt_oracle.c; and through libptmi when a GPU is present) and reports,
for each of the named assumptions the restatement rests on, whether the dump bears it out:

    A1  SFC32's raw step is PractRand's sfc32                         probe_states -> probe_words
    A2  createWith = three-word seeding, counter 1, 15 outputs discarded   words -> created
    A3  the state's planes are (a, b, c, counter) in toVectors order       which permutation of `created` matches
    A4  random @Float = mwc-random's wordToFloat                       probe_states -> probe_floats
    A5  which seed survives `combine` in render Streams                 RNG planes of streams_1 under either rule
    A6  every f32 operation rounded on its own  }                      inline_1 / inline_2 against the oracle: bit-identical, or the
    A7  sin / cos = glibc's sinf / cosf         }                      fraction of pixels within 1e-4 and with an identical RNG state

On a GPU box it asks one question more, the one about A6 only a real dump can answer: libptmi renders the dump's inputs in BOTH arithmetics
-- PTMI_ARITH_EXACT (every operation rounded on its own: the oracle's reading, the default) and PTMI_ARITH_CONTRACTED (a * b + c fused, what
Accelerate's fast-math LLVM backends may do on an FMA host) -- and the report says which one the dump is closer to (`arithmetic`).

Nothing here runs the reference.  `--synthesize OUT [--perturb A1|A2|A3|A4|A5|A6] [--size WxH]` writes a dump in the same format FROM
THE ORACLE (optionally with one assumption deliberately broken; A6: the Inline planes from libptmi's contracted arithmetic, which needs the
GPU): what tests/test_ghc_dump.py feeds back in."""
import argparse
import itertools
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

MAGIC = b"PTMIDUMP"
LIMIT = 15                       # `traceInline 15` (src/Scene/Trace.hs:200)
STREAM_CAP = 1 << 16
U32 = np.uint32


# ---------------------------------------------------------------------------------------------------------------------
# SFC32 in numpy, with the knobs a diagnosis turns (the normative restatement is oracle/pt_oracle.c; test_ghc_dump.py holds the two together)
# ---------------------------------------------------------------------------------------------------------------------
def sfc32_next(state, rot=21, rshift=9, lshift=3):
    """one raw step on arrays (a, b, c, counter): returns (output, new state).  PractRand: rot 21, >> 9, << 3."""
    a, b, c, n = (np.asarray(x, U32) for x in state)
    with np.errstate(over="ignore"):
        out = a + b + n
        n2 = n + U32(1)
        a2 = b ^ (b >> U32(rshift))
        b2 = c + (c << U32(lshift))
        c2 = ((c << U32(rot)) | (c >> U32(32 - rot))) + out
    return out, (a2, b2, c2, n2)


def sfc32_seed3(w0, w1, w2, discard=15, counter=1, **step):
    state = (np.asarray(w0, U32), np.asarray(w1, U32), np.asarray(w2, U32), np.full(np.shape(w0), counter, U32))
    for _ in range(discard):
        _, state = sfc32_next(state, **step)
    return state


def word_to_float(w, variant="mwc"):
    """mwc-random's wordToFloat: (float(int32 w) * 2^-32 + 0.5) + 2^-33, every operation in binary32 -> (0, 1]."""
    w = np.asarray(w, U32)
    if variant == "mwc":
        i = w.view(np.int32).astype(np.float32)
        return (i * np.float32(2.3283064365386963e-10) + np.float32(0.5)) + np.float32(1.1641532182693481e-10)
    if variant == "top24":                                   # (w >> 8) * 2^-24 in [0, 1)
        return (w >> U32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    if variant == "div2^32":                                 # float(w) * 2^-32 in [0, 1]
        return w.astype(np.float32) * np.float32(2.0 ** -32)
    raise ValueError(variant)


def mix32(h):
    h = np.asarray(h, U32).copy()
    with np.errstate(over="ignore"):
        h ^= h >> U32(16); h *= U32(0x85ebca6b); h ^= h >> U32(13); h *= U32(0xc2b2ae35); h ^= h >> U32(16)
    return h


# ---------------------------------------------------------------------------------------------------------------------
# the file
# ---------------------------------------------------------------------------------------------------------------------
def write_dump(path, width, height, sections, limit=LIMIT):
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IIIII", 1, width, height, limit, len(sections)))
        for name, planes in sections:
            f.write(name.encode()[:16].ljust(16, b"\0") + struct.pack("<I", len(planes)))
            for p in planes:
                p = np.ascontiguousarray(p)
                f.write(struct.pack("<IQ", p.dtype.itemsize, p.size) + p.tobytes())


def read_dump(path):
    data = open(path, "rb").read()
    if data[:8] != MAGIC:
        raise ValueError("%s is not a ptmi-dump file" % path)
    version, width, height, limit, n_sections = struct.unpack_from("<IIIII", data, 8)
    if version != 1:
        raise ValueError("dump version %d (this tool reads 1)" % version)
    pos, sections = 28, {}
    for _ in range(n_sections):
        name = data[pos:pos + 16].rstrip(b"\0").decode(); pos += 16
        (n_planes,) = struct.unpack_from("<I", data, pos); pos += 4
        planes = []
        for _ in range(n_planes):
            size, n = struct.unpack_from("<IQ", data, pos); pos += 12
            dtype = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[size]
            planes.append(np.frombuffer(data, dtype, n, pos).copy()); pos += size * n
        sections[name] = planes
    return {"width": width, "height": height, "limit": limit, "sections": sections}


# ---------------------------------------------------------------------------------------------------------------------
# comparisons
# ---------------------------------------------------------------------------------------------------------------------
def first_difference(a, b, width):
    bad = np.flatnonzero(np.asarray(a).reshape(-1) != np.asarray(b).reshape(-1))
    if bad.size == 0:
        return None
    i = int(bad[0])
    return {"pixel_y_x": [i // width, i % width], "index": i, "count": int(bad.size)}


def compare_planes(got, want, width):
    """got / want: 7 planes (r g b as float32 bit patterns or floats, 4 x u32).  NaN payloads are not compared."""
    res = {}
    g = [np.asarray(p).reshape(-1) for p in got]
    w = [np.asarray(p).reshape(-1) for p in want]
    gf = [p.view(np.float32) if p.dtype != np.float32 else p for p in g[:3]]
    wf = [p.view(np.float32) if p.dtype != np.float32 else p for p in w[:3]]
    same_bits = np.ones(g[0].size, bool)
    within = np.ones(g[0].size, bool)
    for a, b in zip(gf, wf):
        eq = (a.view(U32) == b.view(U32)) | (np.isnan(a) & np.isnan(b))
        same_bits &= eq
        with np.errstate(invalid="ignore", divide="ignore"):
            rel = np.abs(a - b) / np.maximum(np.abs(b), np.float32(1e-30))
        within &= eq | (rel <= 1e-4)
    rng_same = np.ones(g[0].size, bool)
    for a, b in zip(g[3:], w[3:]):
        rng_same &= a.view(U32) == b.view(U32)
    n = float(g[0].size)
    res["colour_bit_identical"] = float(same_bits.sum() / n)
    res["colour_within_1e-4"] = float(within.sum() / n)
    res["rng_state_identical"] = float(rng_same.sum() / n)
    bad = np.flatnonzero(~(same_bits & rng_same))
    res["first_differing_pixel_y_x"] = None if bad.size == 0 else [int(bad[0]) // width, int(bad[0]) % width]
    return res


def oracle_steps(pkg, ora, width, height, limit, start, algorithm, seed_rule=None):
    """two successive one-sample calls on the oracle -> (planes after 1, planes after 2)"""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    threads = min(ora.max_threads(), 16)
    outs, state = [], start
    for _ in range(2):
        if algorithm == "inline":
            state = ora.render_inline(sp, pl, cam, width, height, limit, 1, state, n_threads=threads)[0]
        else:
            state = ora.render_streams(sp, pl, cam, width, height, STREAM_CAP, 1, state, seed_rule=seed_rule, n_threads=threads)[0]
        outs.append(state)
    return outs


def device_steps(pkg, width, height, limit, start, algorithm, seed_rule=None, arithmetic=None):
    """the same through libptmi (None when no GPU or library is present); arithmetic: PTMI_OPT_ARITHMETIC for render Inline"""
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        B = pkg.binding
        sp, pl = pkg.world.main_scene()
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.resize(width, height)
            if seed_rule is not None:
                c.set_option(B.OPT_STREAMS_SEED_RULE, seed_rule)
            if arithmetic is not None:
                c.set_option(B.OPT_ARITHMETIC, arithmetic)
            c.upload_state(*start)
            outs = []
            for _ in range(2):
                c.render(pkg.world.initial_camera(), limit, 1, pkg.INLINE if algorithm == "inline" else pkg.STREAMS)
                outs.append(c.download_state())
        return outs
    except Exception as e:        # noqa: BLE001 -- a report must still come out
        return "device path failed: %s" % e


def arithmetic_question(pkg, W, H, limit, start, dump_planes):
    """render Inline of the dump's inputs through libptmi under PTMI_ARITH_EXACT and PTMI_ARITH_CONTRACTED, each against the dump's inline_1 /
    inline_2: {"exact": {...}, "contracted": {...}, "closer_to_the_dump": "exact" | "contracted" | "neither"}, or None without a GPU.  Closer =
    more pixels with an identical RNG state (a branch that went the same way), then more bit-identical colours, then more within 1e-4 -- after
    the last sample the dump holds."""
    B = pkg.binding
    out = {}
    for name, mode in (("exact", B.ARITH_EXACT), ("contracted", B.ARITH_CONTRACTED)):
        dev = device_steps(pkg, W, H, limit, start, "inline", arithmetic=mode)
        if dev is None:
            return None
        if isinstance(dev, str):
            return {"error": dev}
        out[name] = {}
        for k in (1, 2):
            got = dump_planes("inline_%d" % k)
            out[name]["after_%d" % k] = None if got is None else compare_planes(got, dev[k - 1], W)

    def score(entry):
        m = entry.get("after_2") or entry.get("after_1")
        return None if m is None else (m["rng_state_identical"], m["colour_bit_identical"], m["colour_within_1e-4"])
    se, sc = score(out["exact"]), score(out["contracted"])
    if se is None or sc is None:
        out["closer_to_the_dump"] = None
    else:
        out["closer_to_the_dump"] = "exact" if se > sc else ("contracted" if sc > se else "neither")
        out["identical_to_the_dump"] = [n for n, sco in (("exact", se), ("contracted", sc)) if sco == (1.0, 1.0, 1.0)]
    return out


def as_state(planes4, perm, shape):
    return [np.asarray(planes4[perm[k]], U32).reshape(shape) for k in range(4)]


def analyse(dump, use_device=True):
    pkg, ora = graft.load_package(), graft.load_oracle()
    ora.build()
    W, H, limit, S = dump["width"], dump["height"], dump["limit"], dump["sections"]
    report = {"width": W, "height": H, "limit": limit, "assumptions": {}}
    A = report["assumptions"]
    names = "a b c counter".split()

    # ---- A1: the raw step, from the probed states to the raw words.  The probe's planes may be permuted (A3): try all orders.
    probe, words = S.get("probe_states"), S.get("probe_words")
    perm_probe = None
    if probe is None or words is None or len(probe) != 4 or len(words) != 4:
        A["A1"] = {"status": "undetermined", "detail": "the dump holds no probe_states / probe_words with four planes each (no `random @Word32` instance?)"}
    else:
        for perm in itertools.permutations(range(4)):
            state = tuple(probe[perm[k]] for k in range(4))
            ok = True
            for k in range(4):
                out, state = sfc32_next(state)
                ok = ok and np.array_equal(out, words[k])
            if ok:
                perm_probe = perm
                break
        if perm_probe is not None:
            A["A1"] = {"status": "pass", "detail": "four successive raw draws from %d states equal PractRand's sfc32 step" % probe[0].size,
                       "state_plane_order": [names[perm_probe.index(k)] for k in range(4)]}
        else:
            state = tuple(probe[k] for k in range(4))
            out, _ = sfc32_next(state)
            hint = None
            first_draw_ok = any(np.array_equal(sfc32_next(tuple(probe[perm[k]] for k in range(4)))[0], words[0]) for perm in itertools.permutations(range(4)))
            for rot, rs, ls in itertools.product(range(1, 32), range(1, 17), range(1, 9)):       # does a neighbouring variant of the step match all four draws?
                if (rot, rs, ls) == (21, 9, 3) or hint:
                    continue
                if (rs, ls) != (9, 3) and rot != 21:
                    continue                                     # one knob at a time
                for perm in itertools.permutations(range(4)):
                    state, ok = tuple(probe[perm[k]] for k in range(4)), True
                    for k in range(4):
                        o, state = sfc32_next(state, rot=rot, rshift=rs, lshift=ls)
                        ok = ok and np.array_equal(o, words[k])
                    if ok:
                        hint = "all four draws match the step with rotl(c, %d), b >> %d, c << %d (PractRand: 21, 9, 3)" % (rot, rs, ls)
                        break
            if hint is None and first_draw_ok:
                hint = "the FIRST draw is a + b + counter as expected; the state update differs"
            A["A1"] = {"status": "fail", "detail": "raw draws differ from PractRand's sfc32 step (tmp = a + b + counter; counter += 1; a = b ^ (b >> 9); "
                       "b = c + (c << 3); c = rotl(c, 21) + tmp) under every order of the state's planes", "first_state_index": int(np.flatnonzero(out != words[0])[0]) if np.any(out != words[0]) else None,
                       "hint": hint}

    # ---- A2 / A3: createWith.  Which permutation of the dump's planes is (a, b, c, counter) of the three-word seeding?
    perm_created = None
    w, created = S.get("words"), S.get("created")
    if w is None or created is None or len(w) != 3:
        A["A2"] = A["A3"] = {"status": "undetermined", "detail": "the dump holds no words / created sections"}
    elif len(created) != 4:
        A["A2"] = {"status": "undetermined", "detail": "created has %d planes, not 4" % len(created)}
        A["A3"] = {"status": "fail", "detail": "an SFC32 state is %d planes in toVectors, not the 4 words (a, b, c, counter)" % len(created)}
    else:
        want = sfc32_seed3(w[0], w[1], w[2])
        for perm in itertools.permutations(range(4)):
            if all(np.array_equal(created[perm[k]], want[k]) for k in range(4)):
                perm_created = perm
                break
        if perm_created is not None:
            A["A2"] = {"status": "pass", "detail": "createWith (use words) = state (w0, w1, w2), counter 1, 15 outputs discarded, for all %d pixels" % w[0].size}
            order = [names[perm_created.index(k)] for k in range(4)]
            A["A3"] = {"status": "pass" if perm_created == (0, 1, 2, 3) else "fail", "plane_order_in_toVectors": order,
                       "detail": "planes are (a, b, c, counter)" if perm_created == (0, 1, 2, 3) else "planes are %s: change planesOf / fromPlanes in Scene/HIP.hs (and the order libptmi is handed them in)" % order}
        elif A["A1"]["status"] == "fail":
            A["A2"] = A["A3"] = {"status": "undetermined", "detail": "seeding runs the raw step (A1), which already differs"}
        else:
            hint = None
            for discard, counter in itertools.product(range(0, 33), (0, 1)):
                cand = sfc32_seed3(w[0], w[1], w[2], discard=discard, counter=counter)
                for perm in itertools.permutations(range(4)):
                    if all(np.array_equal(created[perm[k]], cand[k]) for k in range(4)):
                        hint = "matches with %d outputs discarded and the counter starting at %d" % (discard, counter)
            A["A2"] = {"status": "fail", "detail": "created differs from the three-word seeding (counter 1, 15 discards) under every plane order", "hint": hint,
                       "first_difference": first_difference(created[0], want[0], W)}
            A["A3"] = {"status": "undetermined", "detail": "depends on A2"}

    # ---- A4: word -> float
    floats = S.get("probe_floats")
    if probe is None or floats is None or len(floats) != 4 or perm_probe is None:
        A["A4"] = {"status": "undetermined", "detail": "needs probe_floats and a passed A1"}
    else:
        def draws(variant):
            state, ok = tuple(probe[perm_probe[k]] for k in range(4)), True
            for k in range(4):
                out, state = sfc32_next(state)
                ok = ok and np.array_equal(word_to_float(out, variant).view(U32), floats[k].view(U32))
            return ok
        if draws("mwc"):
            A["A4"] = {"status": "pass", "detail": "random @Float = (float(int32 w) * 2^-32 + 0.5) + 2^-33, bit for bit, one raw step per draw"}
        else:
            other = [v for v in ("top24", "div2^32") if draws(v)]
            A["A4"] = {"status": "fail", "detail": "random @Float is not mwc-random's wordToFloat of one raw word", "hint": ("it is the `%s` conversion" % other[0]) if other else None}

    # ---- the whole path: render Inline, then Streams under either seed rule
    perm = perm_created if perm_created is not None else (0, 1, 2, 3)
    report["renders"] = {}
    if created is not None and len(created) == 4:
        start = [np.zeros((H, W), np.float32) for _ in range(3)] + as_state(created, perm, (H, W))

        def dump_planes(name):
            sec = S.get(name)
            if sec is None or len(sec) != 7:
                return None
            return [np.asarray(p, U32).view(np.float32).reshape(H, W) for p in sec[:3]] + as_state(sec[3:], perm, (H, W))
        runs = {"inline": (None,), "streams": (ora.SEED_FROM_RESULT, ora.SEED_KEEP_ACCUMULATOR)}
        for alg, rules in runs.items():
            for rule in rules:
                tag = alg if rule is None else "%s_%s" % (alg, "from_result" if rule == ora.SEED_FROM_RESULT else "keep_accumulator")
                want = oracle_steps(pkg, ora, W, H, limit, start, alg, rule)
                entry = {}
                for k in (1, 2):
                    got = dump_planes("%s_%d" % (alg, k))
                    entry["after_%d" % k] = None if got is None else compare_planes(got, want[k - 1], W)
                if use_device:
                    dev = device_steps(pkg, W, H, limit, start, alg, None if rule is None else (1 if rule == ora.SEED_FROM_RESULT else 0))
                    if isinstance(dev, str):
                        entry["device"] = dev
                    elif dev is not None:
                        entry["device_equals_oracle"] = all(compare_planes(d, o, W)["first_differing_pixel_y_x"] is None for d, o in zip(dev, want))
                report["renders"][tag] = entry
        # ---- which arithmetic is the dump's?  (GPU only: the contracted build exists on the device alone)
        report["arithmetic"] = arithmetic_question(pkg, W, H, limit, start, dump_planes) if use_device else None
        inl = report["renders"]["inline"]
        upstream = all(A[k]["status"] == "pass" for k in ("A1", "A2", "A4"))
        if inl["after_1"] is None:
            A["A6"] = A["A7"] = {"status": "undetermined", "detail": "the dump holds no inline_1 section with seven planes"}
        elif not upstream:
            A["A6"] = A["A7"] = {"status": "undetermined", "detail": "the generator differs upstream (A1 / A2 / A4): the paths draw different numbers", "measured": inl}
        elif all(inl["after_%d" % k] and inl["after_%d" % k]["first_differing_pixel_y_x"] is None for k in (1, 2)):
            A["A6"] = A["A7"] = {"status": "pass", "detail": "render Inline: all seven planes bit-identical to the oracle after 1 and 2 samples"}
        else:
            m = inl["after_2"] or inl["after_1"]
            rounding_only = m["rng_state_identical"] >= 0.99 and m["colour_within_1e-4"] >= 0.99
            A["A6"] = A["A7"] = {"status": "fail", "measured": inl,
                                 "detail": ("render Inline differs from the oracle in rounding only -- contraction / reassociation (A6) or another sin / cos (A7): %.4f %% of the pixels within 1e-4, "
                                            "%.4f %% with an identical RNG state (a differing state = a hit / miss or freeze decision that went the other way)" % (100 * m["colour_within_1e-4"], 100 * m["rng_state_identical"]))
                                 if rounding_only else "render Inline differs from the oracle beyond rounding: the restatement of the path itself is wrong somewhere (first differing pixel above)"}
        # A5
        fr, keep = report["renders"]["streams_from_result"], report["renders"]["streams_keep_accumulator"]
        if fr["after_1"] is None:
            A["A5"] = {"status": "undetermined", "detail": "the dump holds no streams_1 section with seven planes"}
        else:
            f1, k1 = fr["after_1"]["rng_state_identical"], keep["after_1"]["rng_state_identical"]
            if f1 >= 0.99 and f1 > k1:
                A["A5"] = {"status": "pass", "detail": "combine = f new old: the RNG planes after render Streams are the result's seed, advanced (PTMI_SEED_FROM_RESULT, the default): %.4f %% of the pixels" % (100 * f1)}
            elif k1 >= 0.99:
                A["A5"] = {"status": "fail", "detail": "combine = f old new: the pixel keeps its seed (PTMI_SEED_KEEP_ACCUMULATOR matches %.4f %% of the pixels, FROM_RESULT %.4f %%): make it the default" % (100 * k1, 100 * f1)}
            else:
                A["A5"] = {"status": "undetermined", "detail": "neither reading matches (from result %.4f %%, keep %.4f %%)" % (100 * f1, 100 * k1)}
    order = ["A1", "A2", "A3", "A4", "A5", "A6", "A7"]
    failed = [k for k in order if A.get(k, {}).get("status") == "fail"]
    report["first_failure"] = failed[0] if failed else None
    report["all_pass"] = all(A.get(k, {}).get("status") == "pass" for k in order)
    return report


# ---------------------------------------------------------------------------------------------------------------------
# a dump in the same format, from the oracle (what Dump.hs would write if A1-A7 hold) -- optionally with one assumption broken
# ---------------------------------------------------------------------------------------------------------------------
def synthesize(path, width, height, perturb=None):
    pkg, ora = graft.load_package(), graft.load_oracle()
    ora.build()
    i = np.arange(width * height, dtype=np.uint64).astype(U32)
    with np.errstate(over="ignore"):
        w = [mix32((U32(3) * i + U32(k)) ^ U32(0x5EED1234)) for k in range(3)]
    step = {"rot": 20} if perturb == "A1" else {}
    created = sfc32_seed3(w[0], w[1], w[2], discard=12 if perturb == "A2" else 15, **step)
    order = (1, 0, 2, 3) if perturb == "A3" else (0, 1, 2, 3)
    probe = tuple(p[:64] for p in created)
    words, floats, state = [], [], probe
    for _ in range(4):
        out, state = sfc32_next(state, **step)
        words.append(out)
        floats.append(word_to_float(out, "top24" if perturb == "A4" else "mwc").view(U32))
    start = [np.zeros((height, width), np.float32) for _ in range(3)] + [p.reshape(height, width) for p in created]
    streams_rule = ora.SEED_KEEP_ACCUMULATOR if perturb == "A5" else ora.SEED_FROM_RESULT
    inline = oracle_steps(pkg, ora, width, height, LIMIT, start, "inline")
    if perturb == "A6":                                      # the Inline planes as a contracting backend would have produced them (libptmi's measurement mode: GPU)
        inline = device_steps(pkg, width, height, LIMIT, start, "inline", arithmetic=pkg.binding.ARITH_CONTRACTED)
        if inline is None or isinstance(inline, str):
            raise SystemExit("--perturb A6 needs a GPU (the contracted arithmetic exists on the device only): %s" % inline)
    streams = oracle_steps(pkg, ora, width, height, LIMIT, start, "streams", streams_rule)

    def seven(planes):
        return [np.asarray(p, np.float32).view(U32).reshape(-1) for p in planes[:3]] + [np.asarray(planes[3 + order[k]], U32).reshape(-1) for k in range(4)]
    write_dump(path, width, height, [
        ("words", w), ("created", [created[order[k]] for k in range(4)]), ("probe_states", [probe[order[k]] for k in range(4)]),
        ("probe_words", words), ("probe_floats", floats),
        ("inline_1", seven(inline[0])), ("inline_2", seven(inline[1])), ("streams_1", seven(streams[0])), ("streams_2", seven(streams[1]))])


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dump", nargs="?")
    ap.add_argument("--json")
    ap.add_argument("--no-device", action="store_true")
    ap.add_argument("--synthesize", metavar="OUT")
    ap.add_argument("--perturb", choices=["A1", "A2", "A3", "A4", "A5", "A6"])
    ap.add_argument("--size", default="96x64")
    args = ap.parse_args()
    if args.synthesize:
        width, height = (int(v) for v in args.size.split("x"))
        synthesize(args.synthesize, width, height, args.perturb)
        print("wrote", args.synthesize)
        return 0
    if not args.dump:
        ap.error("a dump file, or --synthesize OUT")
    report = analyse(read_dump(args.dump), use_device=not args.no_device)
    for key in ("A1", "A2", "A3", "A4", "A5", "A6", "A7"):
        a = report["assumptions"].get(key, {"status": "undetermined", "detail": ""})
        print("%s  %-12s %s%s" % (key, a["status"].upper(), a.get("detail", ""), ("  [hint: %s]" % a["hint"]) if a.get("hint") else ""))
    for tag, entry in report.get("renders", {}).items():
        for k in ("after_1", "after_2"):
            m = entry.get(k)
            if m:
                print("   %-26s %s  colour bit-identical %.6f  within 1e-4 %.6f  RNG state identical %.6f  first differing pixel %s" % (
                    tag, k, m["colour_bit_identical"], m["colour_within_1e-4"], m["rng_state_identical"], m["first_differing_pixel_y_x"]))
        if "device_equals_oracle" in entry:
            print("   %-26s libptmi == oracle on these inputs: %s" % (tag, entry["device_equals_oracle"]))
    ar = report.get("arithmetic")
    if ar and "closer_to_the_dump" in ar:
        for name in ("exact", "contracted"):
            m = ar[name].get("after_2") or ar[name].get("after_1")
            if m:
                print("   libptmi, %-10s arithmetic against the dump's render Inline: colour bit-identical %.6f  within 1e-4 %.6f  RNG state identical %.6f" % (
                    name, m["colour_bit_identical"], m["colour_within_1e-4"], m["rng_state_identical"]))
        print("   the dump's arithmetic is closer to: %s%s" % (ar["closer_to_the_dump"], (" (identical to: %s)" % ", ".join(ar["identical_to_the_dump"])) if ar.get("identical_to_the_dump") else ""))
    elif ar is None:
        print("   (no GPU here: which arithmetic -- exact or contracted -- the dump is closer to is asked on a GPU box only)")
    print("first failure:", report["first_failure"], "| all pass:", report["all_pass"])
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)
    return 0 if report["all_pass"] else 1


if __name__ == "__main__":
    sys.exit(main())
