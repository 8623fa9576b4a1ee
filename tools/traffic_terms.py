#!/usr/bin/env python3
"""traffic_terms.py -- where the HBM bytes of the stream form's split kernel go, term by term (run ON the GPU box, from the repo root):

    python3 tools/traffic_terms.py [--out gpurun_out/traffic_terms] [-- <bench.py workload arguments>]

Default workload: the glass scene at 1080p / 64 spp through the stream form.  For each VARIANT -- the product, option settings that
change how many items a start hit is cut into, and measurement builds that leave one source of traffic out (-DPTMI_TRAFFIC_SKIP,
csrc/ptmi_diag.h: their planes are wrong on purpose) -- the same bench command runs under rocprofv3 three times: kernel trace, --pmc
FETCH_SIZE, --pmc WRITE_SIZE (counters in their own passes, the program itself after `--`).  Per kernel of the call: calls, mean
duration, fetched MB (FETCH_SIZE x 2 KiB: the guide's gfx950 correction) and written MB (WRITE_SIZE KiB) per call; then the differences
against the product, which are the terms:

    product - skip_item_atomics      what the three colour atomics at every item's end cost in fetches and write-backs
    product - skip_ring_atomics      ... and the atomics of rays taken from the ring
    product - skip_item_costs        ... and the cost record of an item (one atomic on a word all XCDs share; pass 0's items only since round 5)
    pass_by_pass - product           what the ticket order saves: PTMI_OPT_STREAM_PASS_GROUPS = 1 is round 4's order, every region's records from HBM once per pass
    one_group, groups_of_2 / 3, short_2 / 3    other ticket orders
    uniform_4x16                     four passes instead of six: what a pass costs
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

DEFAULT_WORKLOAD = ["--scene", "glass", "--algorithm", "streams", "--streams-form", "stream"]
# name -> (extra compile flags of a measurement build or None, bench options)
VARIANTS = collections.OrderedDict([
    ("product", (None, [])),                                                     # tickets in groups of equal-size passes (automatic)
    ("pass_by_pass", (None, ["--option", "STREAM_PASS_GROUPS=1"])),             # round 4's order
    ("short_2", (None, ["--option", "STREAM_PASS_GROUPS=2"])),
    ("short_3", (None, ["--option", "STREAM_PASS_GROUPS=3"])),
    ("groups_of_2", (None, ["--option", "STREAM_PASS_GROUPS=102"])),
    ("groups_of_3", (None, ["--option", "STREAM_PASS_GROUPS=103"])),
    ("one_group", (None, ["--option", "STREAM_PASS_GROUPS=164"])),
    ("uniform_4x16", (None, ["--option", "STREAM_GRADED=0"])),
    ("skip_item_atomics", (["-DPTMI_TRAFFIC_SKIP=1"], [])),
    ("skip_ring_atomics", (["-DPTMI_TRAFFIC_SKIP=2"], [])),
    ("skip_item_costs", (["-DPTMI_TRAFFIC_SKIP=4"], [])),
])


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("ptmi::", "")
    return name.split("(")[0][:60]


def profile(tag, out, bench_args):
    env = dict(os.environ, TMPDIR="/tmp")
    base = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args + ["--steps", "4", "--warmup", "1", "--ramp-spp", "0", "--no-cpu-baseline", "--no-also"]
    runs = {"stats": ["--kernel-trace", "--stats"], "fetch": ["--pmc", "FETCH_SIZE", "--kernel-trace"], "write": ["--pmc", "WRITE_SIZE", "--kernel-trace"]}
    rec = {}
    for what, flags in runs.items():
        d = os.path.join(out, tag, what)
        res = subprocess.run(["rocprofv3"] + flags + ["-d", d, "--output-format", "csv", "--"] + base, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        if res.returncode != 0:
            raise RuntimeError("%s / %s failed:\n%s" % (tag, what, res.stderr[-2000:]))
        if what == "stats":
            rec["_bench"] = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
            for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
                for r in csv.DictReader(open(f)):
                    k = rec.setdefault(short(r["Kernel_Name"]), {"calls": 0, "ns": 0})
                    k["calls"] += 1
                    k["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        else:
            key, scale = ("fetch_MB", 2 * 1024 / 1e6) if what == "fetch" else ("write_MB", 1024 / 1e6)
            n = collections.Counter()
            for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    k = rec.setdefault(short(r["Kernel_Name"]), {"calls": 0, "ns": 0})
                    k[key] = k.get(key, 0.0) + float(r["Counter_Value"]) * scale
                    n[short(r["Kernel_Name"])] += 1
            for name, count in n.items():
                rec[name][key] = round(rec[name][key] / count, 2)
        print("%s: %s done" % (tag, what), file=sys.stderr, flush=True)
    total = sum(v["ns"] for k, v in rec.items() if k != "_bench") or 1
    res = {"kernel_ms_of_the_bench_line": rec["_bench"]["roofline"]["kernel_ms"], "build_id": rec["_bench"].get("binary_build_id"), "kernels": {}}
    for name, v in rec.items():
        if name == "_bench" or not v["calls"] or v["ns"] < 0.002 * total:
            continue
        res["kernels"][name] = {"calls": v["calls"], "avg_us": round(v["ns"] / v["calls"] / 1e3, 1), "fetch_MB_per_call": v.get("fetch_MB"), "write_MB_per_call": v.get("write_MB")}
    per_call = [v for k, v in res["kernels"].items() if "primary" not in k]           # (the start-hit list is built once per camera, not per call)
    res["call_MB"] = {"fetch": round(sum(v["fetch_MB_per_call"] or 0 for v in per_call), 1), "write": round(sum(v["write_MB_per_call"] or 0 for v in per_call), 1)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "traffic_terms"))
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("workload", nargs="*", default=None)
    args = ap.parse_args()
    workload = args.workload or DEFAULT_WORKLOAD
    pkg = graft.load_package()
    pkg._build.build_lib()
    os.makedirs(args.out, exist_ok=True)
    out = {"workload": " ".join(workload), "variants": {}}
    for name in args.variants.split(","):
        flags, options = VARIANTS[name]
        bench_args = list(workload) + options
        if flags:
            lib = os.path.join(ROOT, "build", "libptmi_%s.so" % name)
            pkg._build.build_lib(out=lib, extra_flags=flags)
            bench_args += ["--library", lib]
        out["variants"][name] = profile(name, args.out, bench_args)
    split = next((k for k in out["variants"].get("product", {}).get("kernels", {}) if k.startswith("streams_split_kernel")), None)
    base = out["variants"].get("product", {}).get("kernels", {}).get(split)
    if base:
        out["terms_MB_per_call_of_the_split_kernel"] = {}
        for name, v in out["variants"].items():
            k = v["kernels"].get(split)
            if k and name != "product":
                out["terms_MB_per_call_of_the_split_kernel"]["product - " + name] = {
                    "fetch": round(base["fetch_MB_per_call"] - k["fetch_MB_per_call"], 1), "write": round(base["write_MB_per_call"] - k["write_MB_per_call"], 1),
                    "kernel_us": round(base["avg_us"] - k["avg_us"], 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
