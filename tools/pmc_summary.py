#!/usr/bin/env python3
"""pmc_summary.py DIR -- per-kernel totals of a tools/pmc_kernels.sh run: calls, time (kernel trace), SQ counters summed
over all dispatches of the kernel, and what they imply (active lanes per VALU instruction, SIMD cycles per VALU
instruction at the clock GRBM_GUI_ACTIVE implies, share of wave cycles spent waiting).  All figures are totals over the
profiled run (ramp + warm-up + 4 timed steps of bench.py, every launch at the workload's own sample count: pmc_kernels.sh
passes --ramp-spp 0); n / min / median / max of the dispatch durations say whether the mean is a mean of ONE launch size
(`one_launch_size`: nine dispatches in ten within 20 % of the median; a profile that mixes sizes describes neither), and `_bench` carries the bench line's own
kernel time under the profiler, which tools/valu_roofline.py holds every entry against."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(ptmi::RenderArgs.*", "", name)
    return name.replace("void ptmi::", "").replace("ptmi::", "")[:90]


def main():
    src = sys.argv[1]
    calls, ns = collections.Counter(), collections.Counter()
    lo, hi, every = {}, {}, collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            calls[k] += 1
            ns[k] += d
            lo[k], hi[k] = min(lo.get(k, d), d), max(hi.get(k, d), d)
            every[k].append(d)
    counters = collections.defaultdict(collections.Counter)
    dispatches = collections.defaultdict(collections.Counter)
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dispatches[k][r["Counter_Name"]] += 1
    out = {}
    for k in sorted(ns, key=lambda k: -ns[k]):
        if ns[k] < 0.002 * sum(ns.values()):
            continue
        c = counters.get(k, {})
        rec = {"calls": calls[k], "total_ms": round(ns[k] / 1e6, 3), "avg_us": round(ns[k] / calls[k] / 1e3, 2),
               "min_us": round(lo[k] / 1e3, 2), "max_us": round(hi[k] / 1e3, 2), "median_us": round(sorted(every[k])[len(every[k]) // 2] / 1e3, 2),
               # one size = nine dispatches in ten within 20 % of the median (the first, cold launches of a process run up to 30 % longer)
               "one_launch_size": sum(1 for d in every[k] if abs(d - sorted(every[k])[len(every[k]) // 2]) <= 0.2 * sorted(every[k])[len(every[k]) // 2]) >= 0.9 * len(every[k]),
               "share_of_gpu_time": round(ns[k] / sum(ns.values()), 4)}
        # a PMC pass may see a different number of dispatches than the trace pass (same command, same count expected)
        scale = {name: calls[k] / dispatches[k][name] for name in c if dispatches[k][name]}
        cc = {name: c[name] * scale[name] for name in c}
        rec["counters_total"] = {name: round(v) for name, v in sorted(cc.items())}
        if "SQ_INSTS_VALU" in cc:
            rec["valu_wave_instr_per_call"] = round(cc["SQ_INSTS_VALU"] / calls[k])
            if "SQ_THREAD_CYCLES_VALU" in cc:
                rec["active_lanes_per_valu_instr"] = round(cc["SQ_THREAD_CYCLES_VALU"] / cc["SQ_INSTS_VALU"], 2)
            n_simd, n_xcd = 1024.0, 8.0
            if "GRBM_GUI_ACTIVE" in cc and ns[k]:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs (each counts its own busy cycles): one XCD's share is the launch's
                # duration in shader clocks
                cycles = cc["GRBM_GUI_ACTIVE"] / n_xcd
                rec["clock_ghz_from_GRBM_GUI_ACTIVE"] = round(cycles / ns[k], 3)
                rec["simd_cycles_per_valu_instr"] = round(n_simd * cycles / cc["SQ_INSTS_VALU"], 3)
        if "SQ_WAIT_INST_ANY" in cc and "SQ_WAVE_CYCLES" in cc:
            rec["wave_cycles_waiting_share"] = round(cc["SQ_WAIT_INST_ANY"] / cc["SQ_WAVE_CYCLES"], 3)
        if "FETCH_SIZE" in cc:
            rec["hbm_read_MB_per_call"] = round(cc["FETCH_SIZE"] * 1024 * 2 / calls[k] / 1e6, 2)     # gfx950: x2 (guide)
        if "WRITE_SIZE" in cc:
            rec["hbm_write_MB_per_call"] = round(cc["WRITE_SIZE"] * 1024 / calls[k] / 1e6, 2)
        out[k] = rec
    try:
        bench = json.load(open(os.path.join(src, "bench.json")))
        out["_bench"] = {"workload": bench["config"]["workload"], "ms_per_step_under_rocprof": bench["ms_per_step"],
                         "kernel_ms_under_rocprof": bench["roofline"]["kernel_ms"], "steps": bench["steps"], "warmup": bench["warmup"],
                         "ramp": bench.get("ramp"), "binary_build_id": bench.get("binary_build_id")}
    except Exception:
        pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
