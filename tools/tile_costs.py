#!/usr/bin/env python3
"""tile_costs.py -- how unevenly the work of the glass scene is spread over the 8x8 tiles (the chunks the stream form hands to ONE
wave): per-pixel cost = loop trips of the per-pixel tree walk (-DPTMI_TREE_STATS_MAP writes them into the red plane), 1080p."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "ab", "treemap.so")
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        pkg._build.build_lib(out=out, extra_flags=["-DPTMI_TREE_STATS", "-DPTMI_TREE_STATS_MAP"])
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        return
    pkg.binding._lib = None
    pkg.binding.load_library(out)
    sp, pl = pkg.world.glass_scene()
    w, h, spp = 1920, 1080, 16
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.resize(w, h)
        ctx.set_variant(13)
        ctx.init_output(0x5EED1234)
        ctx.render(pkg.world.initial_camera(), 8, spp, pkg.STREAMS)
        cost = ctx.download_color()[0].astype(np.float64) / spp          # trips per sample
    tiles = cost.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64).sum(1)     # trips per tile and sample (64 pixels)
    order = np.sort(tiles)[::-1]
    total = order.sum()
    cum = np.cumsum(order) / total
    res = {"tiles": int(tiles.size), "mean_trips_per_tile_sample": round(float(tiles.mean()), 1),
           "percentiles": {str(p): round(float(np.percentile(tiles, p)), 1) for p in (50, 90, 99, 99.9, 100)},
           "share_of_work_in_top_tiles": {str(f): round(float(cum[int(tiles.size * f) - 1]), 3) for f in (0.001, 0.01, 0.05, 0.1, 0.25, 0.5)},
           "pixel_max_over_mean": round(float(cost.max() / cost.mean()), 1),
           "wave_trips_for_one_sample_of_the_heaviest_tile": round(float(order[0] / 61.0), 1),
           "wave_trips_per_sample_if_spread_evenly": round(float(total / (6144 * 61.0)), 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
