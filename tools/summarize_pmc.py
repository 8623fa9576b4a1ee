#!/usr/bin/env python3
"""Summarises a directory of rocprofv3 runs (one sub-directory per --pmc pass, as produced by the
command lines in DESIGN.md "how the numbers were taken") into profiles/<tag>_pmc_summary.json and
profiles/traffic.json (HBM bytes per launch of the dominant kernel, corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE x 2 on gfx950, calibrated here on the known 58.06 MB the
kernel reads; WRITE_SIZE as is; both counters are in KiB).
usage: summarize_pmc.py gpurun_out/prof_r01 r01"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    counters, durations = {}, []
    for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render_inline" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            counters[k] = sum(v) / len(v)
    for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if "render_inline" in r["Kernel_Name"]:
                durations.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out = {"kernel": "render_inline_kernel<true, kCached, 8> (C2: 1920x1080, 64 spp, limit 8, S16), one wave per workgroup = an 8x8 pixel tile, 6 waves/SIMD",
           "counters_mean_per_launch": counters}
    if durations:
        out["kernel_trace_ms"] = {"n": len(durations), "mean": sum(durations) / len(durations), "min": min(durations)}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch = counters["FETCH_SIZE"] * 1024 * 2          # gfx950: FETCH_SIZE counts 64 B per 128-B request
        write = counters["WRITE_SIZE"] * 1024
        known = 7 * 1920 * 1080 * 4
        out["hbm"] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "known_read_bytes": known,
                      "known_write_bytes": known, "fetch_calibration": fetch / known, "write_calibration": write / known}
        json.dump({"hbm_bytes_per_launch": round(fetch + write), "source": "profiles/%s_pmc_summary.json" % tag,
                   "workload": "C2"}, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
    if "SQ_INSTS_VALU" in counters:
        live = 1920 * 1080 * 64 * 8 * 0.2815
        out["derived"] = {
            "valu_wave_instr_per_launch": counters["SQ_INSTS_VALU"],
            "avg_active_lanes_per_valu_instr": counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_INSTS_VALU"],
            "valu_lane_instr_per_live_bounce": counters["SQ_THREAD_CYCLES_VALU"] / live,
        }
        if durations:
            simd_cycles = 1024 * out["kernel_trace_ms"]["mean"] * 1e-3 * 2.4e9
            out["derived"]["simd_cycles_per_valu_instr_at_2.4GHz"] = simd_cycles / counters["SQ_INSTS_VALU"]
    json.dump(out, open(os.path.join(root, "profiles", "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
