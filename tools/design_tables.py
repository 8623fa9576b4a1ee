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
    out["build id"] = ("**Which binary.**  Every line above comes from a process whose `libptmi.so` answered `ptmi_build_id()` = `%s`, the hash of the kernel\n"
                       "sources, headers and flags of this commit" % b["binary_build_id"])
    c0, c0b, c3, c4, c4p = also("C0: "), also("C0 at 30"), also("C3"), also("C4: "), also("C4, one part")
    st_c, st_s = also("C2 through render Streams, per-pixel"), also("C2 through render Streams, stream")
    g_t, g_s = also("glass scene, 1920x1080, 64 spp, render Streams, tree"), also("glass scene, 1920x1080, 64 spp, render Streams, stream")
    c5t, c5s = also("C5, one part of 8: "), also("C5, one part of 8, stream")
    cpu = b["cpu_baseline"]
    rows = [
        "| C2 -- this round's run (a %.2f-GHz box) | 1× MI355X | **%s** | %.3f (%.3f) | %s (%.1f %%) | physical HBM %.0f MB per launch = %.0f GB/s (%.1f %%); VALU issue ≥ 91 %% |"
        % (valu["clock_ghz"], th(b["value"]), b["ms_per_step"], r["kernel_ms"], th(r["achieved"]), 100 * r["frac"], r["traffic"] / 1e6, r["physical_GBps"], 100 * r["physical_frac"]),
        "| C2 | oracle port, %d host cores (cgroup quota %d of %d) | %.0f | — | %.1f | single thread %.1f; live fraction %.4f; GPU / port ≈ %s×: a reported baseline, not a quality claim |"
        % (cpu["cores"], cpu["cores"], cpu["affinity_cpus"], cpu["value"], cpu["value"] * 7 / 1e3, cpu["single_thread_value"], cpu["live_fraction"], th(round(b["value"] / cpu["value"], -2))),
        "| C0 800×600, `mainScene`, limit 15, 1 spp per call (the reference's own configuration) | 1× MI355X | — | %.3f (%.3f) | — | %.3f ms at 30 spp per call |"
        % (c0["ms_per_step"], c0["kernel_ms"], c0b["ms_per_step"]),
        "| C3 3840×2160, 256 spp | 1× MI355X | %s | %.1f | %s (%.1f %%) | |" % (th(c3["Msamples_per_s"]), c3["ms_per_step"], th(c3["algorithmic_GBps"]), c3["algorithmic_GBps"] / 80.0),
        "| C4 3840×2160, 1024 spp, whole image / one part of 8 | 1× MI355X | %s / — | %.1f / %.2f | %s (%.1f %%) | §6: every part, three stripe heights |"
        % (th(c4["Msamples_per_s"]), c4["ms_per_step"], c4p["ms_per_step"], th(c4["algorithmic_GBps"]), c4["algorithmic_GBps"] / 80.0),
        "| C2 through `render Streams`: chain / stream form | 1× MI355X | — | %.3f / %.3f | — | driver r4: 4.204 / 4.315; `tools/ab.py` best-of, same box: %s / %s |"
        % (st_c["ms_per_step"], st_s["ms_per_step"], span(ab, "streams"), span(ab, "s16_stream")),
        "| glass scene 1080p / 64 spp: tree walk / stream form | 1× MI355X | — | **%.2f / %.2f** | — | driver r4: 7.608 / 7.604; `tools/ab.py` best-of, same box: %s / %s |"
        % (g_t["ms_per_step"], g_s["ms_per_step"], span(ab, "glass_tree"), span(ab, "glass_stream")),
        "| **C5 per part** (glass, 4K / 512 spp, one of 8): tree walk / stream form | 1× MI355X | — | **%.2f / %.2f** | — | driver r4: 29.40 / 28.73; `tools/ab.py`, same box: %s / %s; §6: every part |"
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
        vrow("`streams_pixels_kernel` + the tail beside it (same, stream form)", s["s16_stream"], False, "%.0f (r04: 310)" % s["s16_stream"]["hbm_MB_per_call"]),
        vrow("`render_streams_tree_kernel<true, 8>` (glass 1080p)", s["glass_tree"], True, th(s["glass_tree"]["hbm_MB_per_call"])),
        vrow("`streams_split_kernel<true, true>` (glass 1080p)", s["glass_stream"], True, "**%s** (r04: 2 169)" % th(s["glass_stream"]["hbm_MB_per_call"])),
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
        brow("C5 (glass, 512 spp, `render Streams`; `profiles/%s_c5_part.json`)" % tag, "tree walk (the default with GLASS)", c5b["forms"]["tree"]["whole"]["best_ms"], c5b["forms"]["tree"]["stripes"]),
        brow("", "stream form", c5b["forms"]["stream"]["whole"]["best_ms"], c5b["forms"]["stream"]["stripes"]),
    ])

    pmc = load(tag, "pmc_glass_stream.json")
    k = lambda n: next(v for name, v in pmc.items() if name.startswith(n))      # noqa: E731
    sp, se, ad = k("streams_split_kernel"), k("streams_slot_seeds_kernel"), k("streams_advance_seeds_kernel")
    fr = sp["hbm_read_MB_per_call"] + se["hbm_read_MB_per_call"] + ad["hbm_read_MB_per_call"]
    wr = sp["hbm_write_MB_per_call"] + se["hbm_write_MB_per_call"] + ad["hbm_write_MB_per_call"]
    out["budget rows"] = "\n".join([
        "| `streams_split_kernel` | %.0f | %.0f | fetched: records 3 x 190 + snapshots 6 x 47 (a snapshot line serves one pass) = 854, spilled children read back 54, tickets, record counts and the colour lines of item ends ~ 35; written: spilled children 54, colour atomics 44, cost atomics ~ 35 (the recording launches' share of the profiled mean; 0 in the steady state), the rest statistics |"
        % (sp["hbm_read_MB_per_call"], sp["hbm_write_MB_per_call"]),
        "| `streams_slot_seeds_kernel` | %.0f | %.0f | 33 MB of seed planes + 12 of keys + counts in; passes x slots x 16 B of snapshots out |" % (se["hbm_read_MB_per_call"], se["hbm_write_MB_per_call"]),
        "| `streams_advance_seeds_kernel` | %.0f | %.0f | the four RNG planes, once each way |" % (ad["hbm_read_MB_per_call"], ad["hbm_write_MB_per_call"]),
        "| **the call** | **%s** | **%.0f** | **%.2f GB** (round 4: 2.78 GB); split kernel alone %s MB = %.1f x the planes (r04: 18.7 x, r03: 12.8 x) at %.2f-%.2f ms (median, mean of the profiled calls) |"
        % (th(fr), wr, (fr + wr) / 1e3, th(sp["hbm_read_MB_per_call"] + sp["hbm_write_MB_per_call"]), (sp["hbm_read_MB_per_call"] + sp["hbm_write_MB_per_call"]) / 116.1,
           sp["median_us"] / 1e3, sp["avg_us"] / 1e3),
    ])
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else "r05"
    check = "--check" in sys.argv
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    new = text
    for name, body in tables(tag).items():
        pat = re.compile(r"(<!-- generated: %s \(tools/design_tables.py\) -->\n)(.*?)(\n<!-- end generated: %s -->)" % (re.escape(name), re.escape(name)), re.S)
        if not pat.search(new):
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
