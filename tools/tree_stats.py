#!/usr/bin/env python3
"""tree_stats.py -- where do the lanes of the per-pixel tree walk (render Streams with GLASS) go?  Builds a
-DPTMI_TREE_STATS copy of libptmi (diagnostic, never the measured library), renders the glass scene once at 1080p / 8 spp
and prints the occupancy of the loop's rounds and the wave-tail factor (lane-trips paid by the waves / lane-trips needed)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    out = "/tmp/libptmi_tree_stats.so"
    pkg._build.build_lib(out=out, extra_flags=["-DPTMI_TREE_STATS"])
    pkg.binding._lib = None
    pkg.binding.load_library(out)
    sp, pl = pkg.world.glass_scene()
    w, h, spp = 1920, 1080, 8
    res = {}
    for variant in (13, 0):                                  # 13: tiles in image order; 0: cost-ordered dispatch (second launch)
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.resize(w, h)
            ctx.set_variant(variant)
            ctx.init_output(0x5EED1234)
            ctx.render(pkg.world.initial_camera(), 8, spp, pkg.STREAMS)
            ctx.synchronize()
            ctx.reset_stats()
            ctx.render(pkg.world.initial_camera(), 8, spp, pkg.STREAMS)
            c = ctx.debug_counters().astype(float)
        need, dead, shade, trace, paid = c[1], c[2], c[3], c[4], c[5]
        res["variant_%d" % variant] = {
            "lane_trips_needed": need, "lane_trips_paid_by_waves": paid, "wave_tail_factor": paid / need,
            "dead_finish_occupancy": dead / need, "shade_occupancy": shade / need, "trace_occupancy": trace / need,
            "shade_occupancy_of_paid": shade / paid, "trace_occupancy_of_paid": trace / paid,
            "trips_per_sample": need / (w * h * spp)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
