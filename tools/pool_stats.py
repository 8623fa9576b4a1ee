#!/usr/bin/env python3
"""pool_stats.py -- diagnostic (-DPTMI_POOL_STATS copy of libptmi, never the measured library): for the pooled
second-shade-round kernels (variants 10/11/12 = 2/4/8 waves per workgroup) prints, per wave, the trips, the trips in
which the wave still had work of its own, the B batches it executed, and where its cycles went."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    out = "/tmp/libptmi_pool_stats.so"
    pkg._build.build_lib(out=out, extra_flags=["-DPTMI_ABLATIONS", "-DPTMI_POOL_STATS"])
    pkg.binding._lib = None
    pkg.binding.load_library(out)
    sp, pl = pkg.world.scene16()
    w, h, spp = 1920, 1080, 64
    for variant in (10, 11, 12):
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.resize(w, h)
            ctx.set_variant(variant)
            ctx.init_output(0x5EED1234)
            ctx.reset_stats()
            ctx.render(pkg.world.initial_camera(), 8, spp)
            raw = ctx.debug_counters().astype("uint32")
        trips, alive, batches, items = (float(raw[k]) for k in (1, 2, 3, 4))
        cyc = [int(raw[8 + 2 * k]) | (int(raw[9 + 2 * k]) << 32) for k in range(5)]
        tot = float(sum(cyc))
        print(json.dumps({"variant": variant, "wave_trips": trips, "trips_with_own_work": alive / trips,
                          "B_batches_per_trip": batches / trips, "items_per_batch": items / max(batches, 1),
                          "cycle_share": {"A": cyc[0] / tot, "post+barrier": cyc[1] / tot, "B": cyc[2] / tot,
                                          "barrier+pickup": cyc[3] / tot, "trace": cyc[4] / tot}}), flush=True)


if __name__ == "__main__":
    main()
