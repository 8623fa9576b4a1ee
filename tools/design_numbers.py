#!/usr/bin/env python3
"""design_numbers.py [TAG] -- the figures DESIGN.md section 7 and README.md quote, read from profiles/<TAG>_* (default r04) and printed as the
rows of the two tables, so that a new profile round's numbers are copied, not retyped.  Prints markdown; changes no file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except FileNotFoundError:
        return None


def thousands(x):
    return "{:,}".format(int(round(x))).replace(",", " ")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    b = load("%s_bench.json" % tag)
    r = b["roofline"]
    by_prefix = lambda p: next(e for e in b["also"] if e["workload"].startswith(p))      # noqa: E731
    print("build id of the profiled binary: %s; clock under the render kernel %.2f GHz" % (load("%s_valu_roofline.json" % tag).get("build_id", load("%s_valu_roofline.json" % tag).get("source_hash", "?")), load("%s_valu_roofline.json" % tag).get("clock_ghz", 0)))
    print()
    print("| Config | Device | Msamples/s | ms per step (kernel) | algorithmic GB/s (% of 8 TB/s) | note |")
    print("| C2 -- this round's run | 1x MI355X | **%s** | %.3f (%.3f) | %s (%.1f %%) | physical HBM %.0f MB per launch = %.0f GB/s (%.1f %%); VALU issue %.3f |"
          % (thousands(b["value"]), b["ms_per_step"], r["kernel_ms"], thousands(r["achieved"]), 100 * r["frac"], r["traffic"] / 1e6, r["physical_GBps"],
             100 * r["physical_frac"], r["valu"]["frac"]))
    print("| C2 | oracle port, %d host cores | %.0f | | | |" % (b["cpu_baseline"]["cores"], b["cpu_baseline"]["value"]))
    c0, c0b = by_prefix("C0: "), by_prefix("C0 at 30")
    print("| C0 | | | %.3f (%.3f) | | %.3f ms at 30 spp per call |" % (c0["ms_per_step"], c0["kernel_ms"], c0b["ms_per_step"]))
    c3 = by_prefix("C3")
    print("| C3 | | %s | %.1f | %s (%.1f %%) | |" % (thousands(c3["Msamples_per_s"]), c3["ms_per_step"], thousands(c3["algorithmic_GBps"]), c3["algorithmic_GBps"] / 80.0))
    c4, c4p = by_prefix("C4: "), by_prefix("C4, one part")
    print("| C4 whole / one part of 8 | | %s / - | %.1f / %.2f | %s (%.1f %%) | |" % (thousands(c4["Msamples_per_s"]), c4["ms_per_step"], c4p["ms_per_step"],
                                                                             thousands(c4["algorithmic_GBps"]), c4["algorithmic_GBps"] / 80.0))
    print("| C2 through Streams: chain / stream form | | | %.3f / %.3f | | |" % (by_prefix("C2 through render Streams, per-pixel")["ms_per_step"], by_prefix("C2 through render Streams, stream")["ms_per_step"]))
    print("| glass 1080p: tree walk / stream form | | | **%.2f / %.2f** | | |" % (by_prefix("glass scene, 1920x1080, 64 spp, render Streams, tree")["ms_per_step"],
                                                                             by_prefix("glass scene, 1920x1080, 64 spp, render Streams, stream")["ms_per_step"]))
    c5t, c5s = by_prefix("C5, one part of 8: "), by_prefix("C5, one part of 8, stream")
    own = [load("%s_bench_c5_part.json" % tag)["ms_per_step"], load("%s_bench_c5_part_stream.json" % tag)["ms_per_step"]]
    print("| C5 per part: tree walk / stream form | | | **%.2f / %.2f** | | its own bench runs %.2f / %.2f |" % (c5t["ms_per_step"], c5s["ms_per_step"], own[0], own[1]))
    print()
    ab = []
    path = os.path.join(ROOT, "profiles", "%s_ab_options.txt" % tag)
    if os.path.exists(path):
        for line in open(path):
            if "{" in line:
                ab.append(json.loads(line[line.index("{"):]))
    if ab:
        keys = [k for k in ab[0] if not k.endswith("dropped") and k != "build_id"]
        print("tools/ab.py, best of 7, the runs of the profile round: " + "; ".join("%s %s" % (k, " / ".join("%.2f" % run[k] for run in ab if k in run)) for k in keys))
        print()
    print("| kernel (workload) | VALU instr per call | issue fraction | active lanes of 64 | SIMD cycles per instr | HBM MB per call |")
    v = load("%s_valu_roofline.json" % tag)
    print("| %s (C2) | %.3f G | **%.3f** / %.3f | %.1f | %.2f | %.0f |" % (v["kernel"], v["valu_wave_instr_per_launch"] / 1e9, v["frac_in_profile"],
                                                                        r["valu"].get("frac_priced_with_measured_opcode_costs", 0), 64 * v["active_lane_frac"],
                                                                        v["measured_simd_cycles_per_instr"], v["hbm_MB_per_call"]))
    s = load("%s_valu_roofline_streams.json" % tag)
    for k, x in s.items():
        if isinstance(x, dict) and "kernel" in x:
            per_call = "%.2f G" % (x["valu_wave_instr_per_launch"] / 1e9)      # (since round 5 every profiled launch is the workload's: C5's too)
            hbm = "%.0f" % x["hbm_MB_per_call"]
            print("| %s (%s) | %s | %.3f | **%.1f** | %.2f | %s |" % (x["kernel"].split(" +")[0], k, per_call, x["frac_in_profile"], 64 * x["active_lane_frac"],
                                                                   x["measured_simd_cycles_per_instr"], hbm))
    print()
    c = load("%s_c4_part.json" % tag)
    print("C4 on one GPU: whole image %.1f ms, slowest part of 8 %.2f ms, part / whole %.4f, predicted 8-GPU speedup %.2f" %
          (c["whole"]["best_ms"], c["best"]["slowest_part_ms"], c["best"]["per_part_over_whole"], c["best"]["predicted_speedup"]))
    c5 = load("%s_c5_part.json" % tag) or {}
    for form, x in c5.get("forms", {}).items():
        print("C5 on one GPU, %s: whole image %.1f ms; %s" % (form, x["whole"]["best_ms"], "; ".join(
            "%s-row stripes: slowest part %.2f ms, imbalance %.3f, predicted 8-GPU speedup %.2f" % (s, y["slowest_part_chunks_0_ms"], y["imbalance_chunks_0"], y["predicted_speedup_8_gpus_chunks_0"])
            for s, y in x["stripes"].items())))
    e = load("%s_split_ends.json" % tag) or {}
    for k, x in e.items():
        print("split kernel's waves, %s passes: first ends at %.3f of the launch, mean %.3f (%.3f ms with the probe)" %
              (k, x["first_wave_ends_at_fraction_of_the_last"], x["mean_end_at_fraction_of_the_last"], x["render_ms"]))


if __name__ == "__main__":
    main()
