#!/usr/bin/env python3
"""design_tables.py [TAG] [--check] -- the measured tables of DESIGN.md (sections 6 and 7), GENERATED from profiles/<TAG>_* between the markers
`<!-- generated: NAME (tools/design_tables.py) -->` ... `<!-- end generated: NAME -->`: the run's rows of the configuration table, the VALU
accounting, the one-GPU part bounds, the byte budget of the GLASS call, and the build id they were measured on.  Rewrites DESIGN.md in place;
with --check it changes nothing and exits 1 if the file is not what the profiles say (tests/test_profile_tools.py runs that)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag, name):
    with open(os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))) as f:
        return json.load(f)


def th(x):
    return "{:,}".format(int(round(x))).replace(",", " ")


def ab_runs(tag):
    runs = []
    path = os.path.join(ROOT, "profiles", "%s_ab_options.txt" % tag)
    if os.path.exists(path):
        for line in open(path):
            if "{" in line:
                runs.append(json.loads(line[line.index("{"):]))
    return runs


def span(runs, key):
    v = sorted(r[key] for r in runs if key in r)
    return "--" if not v else ("%.2f" % v[0] if "%.2f" % v[0] == "%.2f" % v[-1] else "%.2f-%.2f" % (v[0], v[-1]))


def tables(tag):
    b = load(tag, "bench.json")
    r = b["roofline"]
    valu = load(tag, "valu_roofline.json")
    streams = load(tag, "valu_roofline_streams.json")
    also = lambda p: next(e for e in b["also"] if e["workload"].startswith(p))      # noqa: E731
    ab = ab_runs(tag)
    out = {}
    out["build id"] = ("**Which binary.**  Every line above comes from a process whose `libptmi.so` answered `ptmi_build_id()` = `%s`: a hash over the\n"
                       "allocated sections of the objects it was linked from" % b["binary_build_id"])
    c0, c0b, c3, c4, c4p = also("C0: "), also("C0 at 30"), also("C3"), also("C4: "), also("C4, one part")
    cl_ch, cl_cp = also("C0 through the compatible closure, chained"), also("C0 through the compatible closure, copying")
    st_c, st_s = also("C2 through render Streams, per-pixel"), also("C2 through render Streams, stream")
    g_t, g_s = also("glass scene, 1920x1080, 64 spp, render Streams, tree"), also("glass scene, 1920x1080, 64 spp, render Streams, stream")
    c5t, c5s = also("C5, one part of 8: "), also("C5, one part of 8, stream")
    cpu = b["cpu_baseline"]
    rows = [
        "| C2 -- this round's run (a %.2f-GHz box) | 1× MI355X | **%s** | %.3f (%.3f) | %s (%.1f %%) | physical HBM %.0f MB per launch = %.0f GB/s (%.1f %%); VALU issue %.0f %% |"
        % (valu["clock_ghz"], th(b["value"]), b["ms_per_step"], r["kernel_ms"], th(r["achieved"]), 100 * r["frac"], r["traffic"] / 1e6, r["physical_GBps"], 100 * r["physical_frac"], 100 * valu["frac_in_profile"]),
        "| C2 | oracle port, %d host cores (cgroup quota %d of %d) | %.0f | — | %.1f | single thread %.1f; GPU / port ≈ %s×: context, not credit |"
        % (cpu["cores"], cpu["cores"], cpu["affinity_cpus"], cpu["value"], cpu["value"] * 7 / 1e3, cpu["single_thread_value"], th(round(b["value"] / cpu["value"], -2))),
        "| C0 800×600, `mainScene`, limit 15 (the reference's own configuration), resident: 1 / 30 spp per call | 1× MI355X | — | %.3f (%.3f) / %.3f | — | |"
        % (c0["ms_per_step"], c0["kernel_ms"], c0b["ms_per_step"]),
        "| C0 through `compileFor`'s closure, one sample per call: **chained / copying** | 1× MI355X | %s / %s | **%.4f / %.3f** | — | `ptmi_render1_chained` (%d of %d inputs found on the device) / `ptmi_render1`; medians of three blocks |"
        % (th(cl_ch["Msamples_per_s"]), th(cl_cp["Msamples_per_s"]), cl_ch["ms_per_step"], cl_cp["ms_per_step"], cl_ch["chain"]["renders_chained"], cl_ch["chain"]["renders_chained"] + cl_ch["chain"]["renders_uploaded"]),
        "| C3 3840×2160, 256 spp | 1× MI355X | %s | %.1f | %s (%.1f %%) | |" % (th(c3["Msamples_per_s"]), c3["ms_per_step"], th(c3["algorithmic_GBps"]), c3["algorithmic_GBps"] / 80.0),
        "| C4 3840×2160, 1024 spp, whole image / one part of 8 | 1× MI355X | %s / — | %.1f / %.2f | %s (%.1f %%) | §6: every part, three stripe heights |"
        % (th(c4["Msamples_per_s"]), c4["ms_per_step"], c4p["ms_per_step"], th(c4["algorithmic_GBps"]), c4["algorithmic_GBps"] / 80.0),
        "| C2 through `render Streams`: chain / stream form | 1× MI355X | — | %.3f / %.3f | — | round 5: 4.177 / 4.273; `tools/ab.py`: %s / %s |"
        % (st_c["ms_per_step"], st_s["ms_per_step"], span(ab, "streams"), span(ab, "s16_stream")),
        "| glass scene 1080p / 64 spp: tree walk / stream form | 1× MI355X | — | **%.2f / %.2f** | — | round 5: 7.627 / 7.698; `tools/ab.py`: %s / %s |"
        % (g_t["ms_per_step"], g_s["ms_per_step"], span(ab, "glass_tree"), span(ab, "glass_stream")),
        "| **C5 per part** (glass, 4K / 512 spp, one of 8): tree walk / stream form | 1× MI355X | — | **%.2f / %.2f** | — | round 5: 29.47 / 29.49; `tools/ab.py`: %s / %s |"
        % (c5t["ms_per_step"], c5s["ms_per_step"], span(ab, "c5_tree"), span(ab, "c5_stream")),
    ]
    out["run rows"] = "\n".join(rows)

    def vrow(label, a, lanes_bold, hbm, both=None):
        frac = "**%.3f** / %.3f" % (a["frac_in_profile"], both) if both is not None else "%.3f" % a["frac_in_profile"]
        lanes = ("**%.1f**" if lanes_bold else "%.1f") % (64 * a["active_lane_frac"])
        return "| %s | %.2f G | %s | %s | %.2f | %s | %s / %s |" % (label, a["valu_wave_instr_per_launch"] / 1e9, frac, lanes, a["measured_simd_cycles_per_instr"], hbm,
                                                                th(a["kernel_us_in_profile"]), th(a["bench_kernel_us_under_rocprof"]))
    s = streams
    out["valu rows"] = "\n".join([
        vrow("`render_inline_kernel<true, 8>` (C2)", valu, False, "%.0f" % valu["hbm_MB_per_call"], valu["priced_with_measured_rates"]["frac"]).replace("%.2f G" % (valu["valu_wave_instr_per_launch"] / 1e9), "%.3f G" % (valu["valu_wave_instr_per_launch"] / 1e9)),
        vrow("`render_streams_kernel<true, 8>` (C2 through Streams)", s["streams"], False, "%.0f" % s["streams"]["hbm_MB_per_call"]),
        vrow("`streams_pixels_kernel` + the tail beside it (same, stream form)", s["s16_stream"], False, "%.0f" % s["s16_stream"]["hbm_MB_per_call"]),
        vrow("`render_streams_tree_kernel<true, 8>` (glass 1080p)", s["glass_tree"], True, th(s["glass_tree"]["hbm_MB_per_call"])),
        vrow("`streams_split_kernel<true, true>` (glass 1080p)", s["glass_stream"], True, th(s["glass_stream"]["hbm_MB_per_call"])),
        vrow("`render_streams_tree_kernel<true, 8>` (C5 part, 512 spp)", s["c5_tree"], False, th(s["c5_tree"]["hbm_MB_per_call"])),
        vrow("`streams_split_kernel<true, true>` (C5 part, 512 spp)", s["c5_stream"], True, th(s["c5_stream"]["hbm_MB_per_call"])),
    ])

    c4b, c5b = load(tag, "c4_part.json"), load(tag, "c5_part.json")

    def cell(x, best):
        t = "%.2f ms, %.3f, %.2f x" % (x["slowest_part_chunks_0_ms"], x["imbalance_chunks_0"], x["predicted_speedup_8_gpus_chunks_0"])
        return "**%s**" % t if best else t

    def brow(label, form, whole, stripes):
        fastest = min(stripes, key=lambda k: stripes[k]["slowest_part_chunks_0_ms"])
        return "| %s | %s | %.1f ms | %s | %s | %s |" % (label, form, whole, cell(stripes["6"], fastest == "6"), cell(stripes["8"], fastest == "8"), cell(stripes["10"], fastest == "10"))
    out["bound rows"] = "\n".join([
        brow("C4 (S16, 1024 spp, `render Inline`; `profiles/%s_c4_part.json`)" % tag, "--", c4b["whole"]["best_ms"], c4b["stripes"]),
        brow("C5 (glass, 512 spp, `render Streams`; `profiles/%s_c5_part.json`)" % tag, "tree walk (`PTMI_FORM_PIXEL`)", c5b["forms"]["tree"]["whole"]["best_ms"], c5b["forms"]["tree"]["stripes"]),
        brow("", "stream form (what `PTMI_FORM_AUTO` picks for a glass part)", c5b["forms"]["stream"]["whole"]["best_ms"], c5b["forms"]["stream"]["stripes"]),
    ])

    pmc = load(tag, "pmc_glass_stream.json")
    k = lambda n: next(v for name, v in pmc.items() if name.startswith(n))      # noqa: E731
    sp, se, ad = k("streams_split_kernel"), k("streams_slot_seeds_kernel"), k("streams_advance_seeds_kernel")
    fr = sp["hbm_read_MB_per_call"] + se["hbm_read_MB_per_call"] + ad["hbm_read_MB_per_call"]
    wr = sp["hbm_write_MB_per_call"] + se["hbm_write_MB_per_call"] + ad["hbm_write_MB_per_call"]
    out["budget rows"] = "\n".join([
        "| `streams_split_kernel` | %.0f | %.0f | records and snapshots once per GROUP of equal-size passes, spilled children there and back, colour atomics; cost atomics only in recording launches |"
        % (sp["hbm_read_MB_per_call"], sp["hbm_write_MB_per_call"]),
        "| `streams_slot_seeds_kernel` | %.0f | %.0f | seed planes + 4-byte slot keys in; passes × slots × 16 B of snapshots out |" % (se["hbm_read_MB_per_call"], se["hbm_write_MB_per_call"]),
        "| `streams_advance_seeds_kernel` | %.0f | %.0f | the four RNG planes, once each way |" % (ad["hbm_read_MB_per_call"], ad["hbm_write_MB_per_call"]),
        "| **the call** | **%s** | **%.0f** | **%.2f GB** (round 4: 2.78); split kernel alone %s MB = %.1f × the planes at %.2f-%.2f ms (median, mean of the profiled calls) |"
        % (th(fr), wr, (fr + wr) / 1e3, th(sp["hbm_read_MB_per_call"] + sp["hbm_write_MB_per_call"]), (sp["hbm_read_MB_per_call"] + sp["hbm_write_MB_per_call"]) / 116.1,
           sp["median_us"] / 1e3, sp["avg_us"] / 1e3),
    ])
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r06"
    check = "--check" in sys.argv
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    new = text
    for name, body in tables(tag).items():
        pat = re.compile(r"(<!-- generated: %s \(tools/design_tables.py\) -->\n)(.*?)(\n<!-- end generated: %s -->)" % (re.escape(name), re.escape(name)), re.S)
        if not pat.search(new):
            if name == "budget rows":                    # (round 6's DESIGN.md refers to HISTORY.md and the profile for the byte budget)
                continue
            print("DESIGN.md has no markers for `%s`" % name, file=sys.stderr)
            return 2
        new = pat.sub(lambda m: m.group(1) + body + m.group(3), new)
    if check:
        if new != text:
            print("DESIGN.md's generated tables are not what profiles/%s_* say: run python tools/design_tables.py %s" % (tag, tag), file=sys.stderr)
            return 1
        return 0
    if new != text:
        open(path, "w").write(new)
        print("DESIGN.md: generated tables rewritten from profiles/%s_*" % tag)
    else:
        print("DESIGN.md: generated tables already current")
    return 0


if __name__ == "__main__":
    sys.exit(main())
