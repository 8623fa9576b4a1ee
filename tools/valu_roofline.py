#!/usr/bin/env python3
"""valu_roofline.py PMC_SUMMARY VALU_RATES [TAG] -- the VALU issue roofline of the C2 launch (profiles/<TAG>_valu_roofline.json);
valu_roofline.py streams TAG -- the same accounting for the Streams kernels (profiles/<TAG>_valu_roofline_streams.json).

The kernel's bound is VALU issue, not HBM (its physical traffic is 0.4 % of the HBM peak).  A SIMD issues one wave64
vector instruction per 2 cycles at best (32 lanes per cycle), one per 4 for half-rate instructions (v_fma_f32, every
f64 operation, conversions, compares and selects, packed f32) and one per 8 for transcendentals -- the classes
tools/valu_rates.hip measures on this chip.  The hardware counts the dynamic instruction mix by category
(SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _F64, CVT, INT32, INT64); whatever it does not categorise (moves, compares,
selects, min/max, lane operations) is priced at the CHEAPEST class, and so is INT32 (which mixes 2- and 4-cycle
opcodes), so that

    min_issue_cycles = sum over categories of count x class cost        is a LOWER bound on the issue cycles the
                                                                        instructions of one launch need, and
    frac = min_issue_cycles / (SIMDs x cycles the launch took)          a lower bound on the VALU issue utilisation.

`priced_with_measured_rates` repeats the sum with the per-opcode costs valu_rates measured at 8 waves per SIMD
(2.2 / 3.6-4.1 / 8.1 cycles: the microbenchmark's own loop control and issue bubbles are inside those figures, so this
version errs high)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
NOMINAL = {"ADD_F32": 2, "MUL_F32": 2, "FMA_F32": 4, "TRANS_F32": 8, "CVT": 4, "INT32": 2, "INT64": 4,
           "ADD_F64": 4, "MUL_F64": 4, "FMA_F64": 4, "TRANS_F64": 16, "OTHER": 2}
MEASURED_OP = {"ADD_F32": "v_add_f32", "MUL_F32": "v_mul_f32", "FMA_F32": "v_fma_f32", "TRANS_F32": "v_sqrt_f32",
               "CVT": "v_cvt_f64_f32 / v_cvt_f32_f64", "INT32": "v_add_u32", "INT64": "v_lshl_add_u32",
               "ADD_F64": "v_add_f64", "MUL_F64": "v_mul_f64", "FMA_F64": "v_fma_f64", "TRANS_F64": "v_sqrt_f32", "OTHER": "v_mov_b32"}


class NotTheWorkload(ValueError):
    pass


def check_against_bench(name, rec, bench, tolerance=0.10):
    """A per-call mean of a profile is evidence for a workload only if the profiled calls WERE that workload: the kernel's mean
    duration in the trace must lie within `tolerance` of the bench line's own kernel time in the same (profiled) process, and the
    dispatches must be of one size.  (Round 4's C5 entries failed both: 72 of 77 dispatches were 64-spp ramp launches, mean 5.6 ms
    against a 28.7-ms step.)  Raises NotTheWorkload; the entry is then left out."""
    if not bench or not bench.get("kernel_ms_under_rocprof"):
        raise NotTheWorkload("%s: the summary has no _bench.kernel_ms_under_rocprof to hold the profile against (re-run tools/pmc_kernels.sh)" % name)
    want_us = bench["kernel_ms_under_rocprof"] * 1e3
    if abs(rec["avg_us"] - want_us) > tolerance * want_us:
        raise NotTheWorkload("%s: %.1f us per call in the profile, %.1f us per step in its own bench line (n %s, min %s, max %s): not a profile of this workload"
                             % (name, rec["avg_us"], want_us, rec.get("calls"), rec.get("min_us"), rec.get("max_us")))
    if rec.get("one_launch_size") is False:
        raise NotTheWorkload("%s: dispatches between %s and %s us: the profile mixes launch sizes" % (name, rec.get("min_us"), rec.get("max_us")))


def account(name, rec, ops8):
    """The VALU issue accounting of one kernel of a pmc_summary (see the module docstring)."""
    c, calls = rec["counters_total"], rec["calls"]
    per = {k: v / calls for k, v in c.items()}
    mix = {k.replace("SQ_INSTS_VALU_", ""): per[k] for k in per if k.startswith("SQ_INSTS_VALU_")}
    total = per["SQ_INSTS_VALU"]
    mix["OTHER"] = total - sum(mix.values())
    nominal = sum(mix[k] * NOMINAL[k] for k in mix)
    # the one class whose price is in doubt: v_fma_f32 -- 2 cycles in the microarchitecture guide's table, 3.6 measured here
    # (build/valu_rates); NOMINAL takes 4 (the class of the other half-rate instructions)
    fma2 = nominal - mix.get("FMA_F32", 0.0) * (NOMINAL["FMA_F32"] - 2.0)
    fma36 = nominal - mix.get("FMA_F32", 0.0) * (NOMINAL["FMA_F32"] - ops8["v_fma_f32"]["cycles"])
    measured_price = sum(mix[k] * ops8[MEASURED_OP[k]]["cycles"] for k in mix)
    n_simds = 1024
    cycles_per_xcd = per["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
    kernel_us = rec["avg_us"]
    clock_ghz = cycles_per_xcd / (kernel_us * 1e3)
    measured_cycles = n_simds * cycles_per_xcd
    return {
        "kernel": name,
        "n_simds": n_simds, "clock_ghz": round(clock_ghz, 4), "kernel_us_in_profile": kernel_us,
        "dispatches_in_profile": {"n": calls, "min_us": rec.get("min_us"), "median_us": rec.get("median_us"), "max_us": rec.get("max_us")},
        "valu_wave_instr_per_launch": total,
        "mix_wave_instr_per_launch": {k: round(v) for k, v in sorted(mix.items())},
        "class_cost_cycles": NOMINAL,
        "min_issue_cycles": nominal, "measured_cycles_in_profile": measured_cycles,
        "frac_in_profile": round(nominal / measured_cycles, 4),
        "frac_with_v_fma_f32_at_2_cycles": round(fma2 / measured_cycles, 4),
        "frac_with_v_fma_f32_as_measured": round(fma36 / measured_cycles, 4), "v_fma_f32_cycles_measured": ops8["v_fma_f32"]["cycles"],
        "avg_issue_cycles_per_instr": round(nominal / total, 4),
        "measured_simd_cycles_per_instr": round(measured_cycles / total, 4),
        "priced_with_measured_rates": {"issue_cycles": measured_price, "frac": round(measured_price / measured_cycles, 4),
                                       "avg_cycles_per_instr": round(measured_price / total, 4),
                                       "rates": {k: ops8[MEASURED_OP[k]]["cycles"] for k in mix}},
        "table_min_max_cycles": [min(v["cycles"] for k, v in ops8.items() if k.startswith("v_")),
                                 max(v["cycles"] for k, v in ops8.items() if k.startswith("v_"))],
        "active_lane_frac": round(per["SQ_THREAD_CYCLES_VALU"] / total / 64.0, 4),
        "wave_cycles_waiting_for_issue_share": rec.get("wave_cycles_waiting_share"),
        "hbm_MB_per_call": round(rec.get("hbm_read_MB_per_call", 0.0) + rec.get("hbm_write_MB_per_call", 0.0), 1) if "hbm_read_MB_per_call" in rec else None,
    }


# the Streams kernels: which kernel of which PMC summary (profiles/<tag>_pmc_<key>.json) stands for which workload of bench.py's `also` block
STREAMS = {"streams": "render_streams_kernel", "s16_stream": "streams_pixels_kernel", "glass_tree": "render_streams_tree_kernel",
           "glass_stream": "streams_split_kernel", "c5_tree": "render_streams_tree_kernel", "c5_stream": "streams_split_kernel"}


def streams(tag):
    """valu_roofline.py streams TAG -> profiles/<TAG>_valu_roofline_streams.json: the same accounting for the Streams kernels."""
    rates = json.load(open(os.path.join(ROOT, "profiles", "%s_valu_rates.json" % tag)))
    ops8 = rates["results"]["waves_per_simd_8"]["ops"]
    fallback_id = os.environ.get("PTMI_PROFILE_BUILD_ID") or graft.load_package()._build.code_id()
    out = {}
    for key, kernel in STREAMS.items():
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, key))
        if not os.path.exists(path):
            continue
        summary = json.load(open(path))
        found = [(k, v) for k, v in summary.items() if kernel in k and "counters_total" in v and "SQ_INSTS_VALU" in v["counters_total"]]
        if not found:
            continue
        name, rec = max(found, key=lambda kv: kv[1]["total_ms"])
        if key == "s16_stream":
            # Two kernels share the chip in this call: the persistent streams_pixels_kernel and, beside it on a low-priority stream, the
            # per-pixel chain kernel that renders the cheap end of the dispatch order (the TAIL).  Neither kernel's own duration is the
            # time its instructions had the SIMDs to themselves, so the accounting is of the CALL: both kernels' instructions over the
            # span from their common start to the end of the later one (the tail kernel's duration: it is enqueued at the fork).
            tail = [(k, v) for k, v in summary.items() if "render_streams_kernel" in k and "counters_total" in v and "SQ_INSTS_VALU" in v["counters_total"]]
            if tail:
                tname, trec = tail[0]
                both = {"calls": rec["calls"], "avg_us": max(rec["avg_us"], trec["avg_us"]), "counters_total": {},
                        "min_us": max(rec.get("min_us", 0), trec.get("min_us", 0)), "max_us": max(rec.get("max_us", 0), trec.get("max_us", 0)),
                        "one_launch_size": rec.get("one_launch_size") and trec.get("one_launch_size")}
                for cname, value in rec["counters_total"].items():
                    both["counters_total"][cname] = value + trec["counters_total"].get(cname, 0.0) * rec["calls"] / trec["calls"]
                span_cycles = rec["counters_total"]["GRBM_GUI_ACTIVE"] / rec["avg_us"] * both["avg_us"]     # the same clock over the longer span
                both["counters_total"]["GRBM_GUI_ACTIVE"] = span_cycles
                for k2 in ("hbm_read_MB_per_call", "hbm_write_MB_per_call"):
                    if k2 in rec:
                        both[k2] = rec[k2] + trec.get(k2, 0.0)
                name, rec = name + " + " + tname + " (the tail beside it)", both
        bench = summary.get("_bench", {})
        try:
            check_against_bench(name, rec, bench)
        except NotTheWorkload as e:
            # a refused workload stays in the file, with the reason: bench.py leaves it out of `also[*].roofline`, and a reader of the
            # tracked profile sees why (it used to vanish without a trace)
            print("REFUSED %s: %s" % (key, e), file=sys.stderr)
            out[key] = {"refused": str(e), "kernel": name, "source": "profiles/%s_pmc_%s.json" % (tag, key)}
            continue
        a = account(name, rec, ops8)
        a.update({"workload": bench.get("workload"), "source": "profiles/%s_pmc_%s.json + profiles/%s_valu_rates.json" % (tag, key, tag),
                  "bench_kernel_us_under_rocprof": round(bench["kernel_ms_under_rocprof"] * 1e3, 1),
                  # the id the PROFILED binary carried (its bench line printed ptmi_build_id()); else what the profile round recorded
                  "build_id": (bench.get("binary_build_id") or fallback_id).split("+")[0]})
        out[key] = a
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_valu_roofline_streams.json" % tag), "w"), indent=1)
    for key, a in out.items():
        if "refused" in a:
            continue
        print("%-13s %-40s issue frac %.3f  active lanes %.3f  cycles / instr %.3f  HBM MB / call %s" % (
            key, a["kernel"][:40], a["frac_in_profile"], a["active_lane_frac"], a["measured_simd_cycles_per_instr"], a["hbm_MB_per_call"]))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "streams":
        return streams(sys.argv[2])
    summary = json.load(open(sys.argv[1]))
    rates = json.load(open(sys.argv[2]))
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    name, rec = next((k, v) for k, v in summary.items() if "render_inline_kernel" in k)
    ops8 = rates["results"]["waves_per_simd_8"]["ops"]
    bench = summary.get("_bench", {})
    check_against_bench(name, rec, bench)                # raises: a C2 profile that is not C2 must not become roofline.valu
    out = account(name, rec, ops8)
    out.update({"workload": bench.get("workload"),
                "source": "rocprofv3 --pmc passes of tools/pmc_kernels.sh c2 (means over all launches of the run) + build/valu_rates",
                "bench_kernel_us_under_rocprof": round(bench["kernel_ms_under_rocprof"] * 1e3, 1),
                # the id the PROFILED binary carried (its bench line printed ptmi_build_id()); else what tools/profile_round.sh recorded
                "build_id": (bench.get("binary_build_id") or os.environ.get("PTMI_PROFILE_BUILD_ID") or graft.load_package()._build.code_id()).split("+")[0]})
    path = os.path.join(ROOT, "profiles", "%s_valu_roofline.json" % tag)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
