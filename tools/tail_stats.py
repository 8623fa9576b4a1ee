#!/usr/bin/env python3
"""tail_stats.py [s16|glass] -- how much of a stream-form launch its end costs (diagnostic build -DPTMI_TAIL_STATS of the
pixels kernel): every persistent wave stamps its start and end (s_memtime), counts its loop trips and the lanes that held an
item in each.  Prints the waves' mean duration against the longest (all waves of the persistent grid start together, so what the mean
lacks to the longest is wave-time after waves have ended), their histogram, and the mean number of lanes with an item per trip."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    phases = "phases" in sys.argv[1:]      # ... and the cycles its waves spend in the refill block and in the trips that end items
    lib = os.path.join(ROOT, "build", "ab", "tailphases.so" if phases else "tailstats.so")
    if not os.path.exists(lib):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        pkg._build.build_lib(out=lib, extra_flags=["-DPTMI_TAIL_STATS"] + (["-DPTMI_TAIL_PHASES"] if phases else []))
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return
    pkg.binding._lib = None
    pkg.binding.load_library(lib)
    B = pkg.binding
    cam = pkg.world.initial_camera()
    sp, pl = pkg.world.scene16()
    out = {}
    for spp in (64,):
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.resize(1920, 1080)
            c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            c.init_output(0x5EED1234)
            for _ in range(40):
                c.render(cam, 8, spp, pkg.STREAMS)
            c.synchronize()
            c.reset_stats()
            c.set_timing(True)
            c.render(cam, 8, spp, pkg.STREAMS)
            ms = c.stats()["last_render_ms"]
            w = c.debug_counters()
            u64 = lambda i: int(w[i]) | (int(w[i + 1]) << 32)    # noqa: E731
            total, waves, lane_trips, trips, longest = u64(12), u64(14), u64(16), u64(18), u64(20)
            out["spp_%d" % spp] = {"render_ms": round(ms, 3), "waves": waves, "mean_wave_cycles": round(total / max(waves, 1)),
                                   "longest_wave_cycles": longest, "mean_over_longest": round(total / max(waves, 1) / max(longest, 1), 4),
                                   "lanes_with_an_item_per_trip": round(lane_trips / max(trips, 1), 2), "trips_per_wave": round(trips / max(waves, 1), 1),
                                   "waves_by_duration_bins_of_2^19_cycles": [int(x) for x in w[24:64]]}
            if phases:
                del out["spp_%d" % spp]["waves_by_duration_bins_of_2^19_cycles"]
                refill, over, refills, ends, taken = u64(24), u64(26), u64(28), u64(30), u64(32)
                out["spp_%d" % spp].update({"refill_share_of_wave_cycles": round(refill / max(total, 1), 4), "item_end_trips_share": round(over / max(total, 1), 4),
                                            "refills_per_wave": round(refills / max(waves, 1), 1), "cycles_per_refill": round(refill / max(refills, 1)),
                                            "items_per_refill": round(taken / max(refills, 1), 2), "item_end_trips_per_wave": round(ends / max(waves, 1), 1),
                                            "cycles_per_item_end_block": round(over / max(ends, 1)), "cycles_per_trip": round(total / max(trips, 1))})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
