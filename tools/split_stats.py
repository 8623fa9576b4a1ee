#!/usr/bin/env python3
"""split_stats.py -- where the lanes of streams_split_kernel go (diagnostic build -DPTMI_SPLIT_STATS): lane participation in
every block of the loop, per wave-trip, on the glass scene at 1080p / 64 spp; and how the waves' durations spread."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


ENDS_FIRST = 160          # 41-us bins: the window is 6.55 .. 9.2 ms after the first wave's start (the launch takes ~8.2 ms)


def ends(pkg):
    """split_stats.py ends -- only WHEN the waves of the split kernel end (-DPTMI_SPLIT_ENDS with 41-us bins: two atomics per wave, the kernel
    otherwise the product's), glass scene, 1080p / 64 spp: does the launch have a tail?"""
    lib = os.path.join(ROOT, "build", "ab", "splitends.so")
    if not os.path.exists(lib):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        pkg._build.build_lib(out=lib, extra_flags=["-DPTMI_SPLIT_ENDS", "-DPTMI_SPLIT_HIST_SHIFT=12", "-DPTMI_SPLIT_HIST_FIRST=%dull" % ENDS_FIRST])
    if "build" in sys.argv:
        return
    pkg.binding._lib = None
    pkg.binding.load_library(lib)
    B = pkg.binding
    sp, pl = pkg.world.glass_scene()
    out = {}
    for graded in (1, 0):
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.resize(1920, 1080)
            c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            c.set_option(B.OPT_STREAM_GRADED, graded)
            c.init_output(0x5EED1234)
            for _ in range(6):
                c.render(pkg.world.initial_camera(), 8, 64, pkg.STREAMS)
            c.synchronize()
            c.reset_stats()
            c.set_timing(True)
            c.render(pkg.world.initial_camera(), 8, 64, pkg.STREAMS)
            ms = c.stats()["last_render_ms"]
            wc = [int(x) for x in c.debug_counters()]
        hist = {ENDS_FIRST + i: n for i, n in enumerate(wc[96:160]) if n}
        first, last = min(hist), max(hist)
        out["graded" if graded else "uniform"] = {"render_ms": round(ms, 3), "bin_us": 40.96, "waves": sum(hist.values()), "waves_ending_per_bin": hist,
                                                  "first_wave_ends_at_fraction_of_the_last": round((first + 0.5) / (last + 0.5), 3),
                                                  "mean_end_at_fraction_of_the_last": round(sum((b + 0.5) * n for b, n in hist.items()) / sum(hist.values()) / (last + 0.5), 3)}
    print(json.dumps(out))


def main():
    pkg = graft.load_package()
    if len(sys.argv) > 1 and sys.argv[1] == "ends":
        return ends(pkg)
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]          # e.g. an experiment's flag beside the statistics
    lib = os.path.join(ROOT, "build", "ab", "splitstats%s.so" % "".join(e[2:].replace("=", "_") for e in extra))
    if not os.path.exists(lib):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        pkg._build.build_lib(out=lib, extra_flags=["-DPTMI_SPLIT_STATS"] + extra)
    if "build" in sys.argv[1:]:
        return
    pkg.binding._lib = None
    pkg.binding.load_library(lib)
    B = pkg.binding
    cam = pkg.world.initial_camera()
    sp, pl = pkg.world.glass_scene()
    out = {}
    for w, h, spp, parts in ((1920, 1080, 64, 1), (3840, 2160, 512, 8)):
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            if parts > 1:
                c.set_partition(10, parts, 0)
            c.resize(w, h)
            c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            c.init_output(0x5EED1234)
            for _ in range(6):
                c.render(cam, 8, spp, pkg.STREAMS)
            c.synchronize()
            c.reset_stats()
            c.set_timing(True)
            c.render(cam, 8, spp, pkg.STREAMS)
            st = c.stats()
            wc = [int(x) for x in c.debug_counters()]
            trips = max(wc[1], 1)
            waves = max(wc[16], 1)
            total = wc[12] | (wc[13] << 32)
            longest = wc[14] | (wc[15] << 32)
            px = c.local_rows * w
            out["%dx%d_%dspp_part_of_%d" % (w, h, spp, parts)] = {
                "render_ms": round(st["last_render_ms"], 3), "wave_trips": wc[1], "wave_trips_per_pixel_sample": round(wc[1] / (px * spp), 3),
                "per_trip": {"lanes_with_an_item": round(wc[11] / trips, 2), "dead_hits": round(wc[2] / trips, 2), "free_lanes": round(wc[3] / trips, 2),
                             "taken_from_ring": round(wc[4] / trips, 2), "samples_started": round(wc[5] / trips, 2), "items_ended": round(wc[6] / trips, 3),
                             "refill_blocks": round(wc[7] / trips, 3), "shades": round(wc[8] / trips, 2), "of_them_glass": round(wc[9] / trips, 2),
                             "traces": round(wc[10] / trips, 2)},
                "waves": waves, "mean_wave_cycles": round(total / waves), "longest_wave_cycles": longest,
                "mean_over_longest": round(total / waves / max(longest, 1), 4), "spilled": st["stream_rays_spilled"], "live": st["live_bounces"],
                "per_xcd": {"waves": wc[24:32], "mean_wave_kcycles": [round(wc[32 + x] * 4.096 / max(wc[24 + x], 1)) for x in range(8)],
                            "longest_wave_kcycles": [round(wc[40 + x] * 4.096) for x in range(8)], "trips_per_wave": [round(wc[48 + x] / max(wc[24 + x], 1)) for x in range(8)]},
                "share_of_wave_time_by_block": dict(zip(["dead hits", "refill", "next ray / sample start / item end", "shade: draws, mirror direction", "GLASS block + expand",
                                                         "shade: rotation, child", "trace", "loop control"],
                                                        [round((wc[224 + 2 * k] | (wc[225 + 2 * k] << 32)) / max(sum(wc[224 + 2 * j] | (wc[225 + 2 * j] << 32) for j in range(8)), 1), 4) for k in range(8)])),
                "when_the_tickets_ran_out_per_wave": {"lanes_with_an_item": round(wc[56] / waves, 1), "their_samples_left": round(wc[57] / waves, 1),
                                                      "spill_records": round(wc[58] / waves, 1), "ring_records": round(wc[59] / waves, 1),
                                                      "trips_after_that": round(wc[60] / waves, 1), "most_trips_after_that": wc[61]},
                "waves_starting_per_164us_bin": wc[64:96], "waves_ending_per_164us_bin_from_32": wc[96:160], "their_mean_trips": [round(t / max(n, 1)) for n, t in zip(wc[96:160], wc[160:224])]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
