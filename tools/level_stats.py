#!/usr/bin/env python3
"""level_stats.py -- lane occupancy of the stream form's level kernels on the glass scene (diagnostic build
-DPTMI_LEVEL_STATS): per wave-trip, how many lanes finish a dead hit, fetch their next piece of work, shade (and how many of
those at GLASS), trace; and how many trips run the refill block."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    out = "/tmp/libptmi_level_stats.so"
    pkg._build.build_lib(out=out, extra_flags=["-DPTMI_LEVEL_STATS"])
    pkg.binding._lib = None
    pkg.binding.load_library(out)
    res = {}
    for scene in ("glass", "s16"):
        sp, pl = {"glass": pkg.world.glass_scene, "s16": pkg.world.scene16}[scene]()
        w, h, spp = 1920, 1080, 64
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.resize(w, h)
            ctx.set_option(pkg.binding.OPT_STREAMS_FORM, pkg.binding.FORM_STREAM)
            ctx.init_output(0x5EED1234)
            ctx.render(pkg.world.initial_camera(), 8, spp, pkg.STREAMS)
            ctx.synchronize()
            ctx.reset_stats()
            ctx.render(pkg.world.initial_camera(), 8, spp, pkg.STREAMS)
            c = ctx.debug_counters().astype(float)
        for k, name in ((0, "level_0"), (1, "levels_1_up")):
            trips, dead, nxt, shade, glass, trace, refills = c[20 + 8 * k: 27 + 8 * k]
            if trips == 0:
                continue
            res["%s_%s" % (scene, name)] = {
                "wave_trips": trips, "wave_trips_per_pixel_sample": trips * 64 / (w * h * spp),
                "lanes_finishing_a_dead_hit": dead / trips, "lanes_fetching_next_work": nxt / trips,
                "lanes_shading": shade / trips, "of_them_at_glass": glass / max(shade, 1), "lanes_tracing": trace / trips,
                "trips_with_a_refill": refills / trips}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
