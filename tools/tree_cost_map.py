#!/usr/bin/env python3
"""tree_cost_map.py -- per-pixel cost (loop trips) of the per-pixel tree walk on the glass scene, from a diagnostic build
(-DPTMI_TREE_STATS_MAP writes the trips into the red plane), and what lane assignments other than 8x8 tiles would pay:
wave cost = its slowest lane; the tail factor is sum(wave cost x 64) / sum(pixel cost)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def tail(cost_by_wave):
    c = cost_by_wave.reshape(-1, 64)
    return float(c.max(1).sum() * 64 / c.sum())


def main():
    pkg = graft.load_package()
    out = "/tmp/libptmi_tree_map.so"
    pkg._build.build_lib(out=out, extra_flags=["-DPTMI_TREE_STATS", "-DPTMI_TREE_STATS_MAP"])
    pkg.binding._lib = None
    pkg.binding.load_library(out)
    scene = sys.argv[1] if len(sys.argv) > 1 else "glass"
    sp, pl = {"glass": pkg.world.glass_scene, "s16": pkg.world.scene16}[scene]()
    w, h = 1920, 1080
    res = {}
    maps = {}
    for spp in (8, 64):
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.resize(w, h)
            ctx.set_variant(13)
            ctx.init_output(0x5EED1234 + spp)
            ctx.render(pkg.world.initial_camera(), 8, spp, pkg.STREAMS)
            cost = ctx.download_color()[0].astype(np.float64)
        maps[spp] = cost
        tiles = cost.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        r = {"tiles_8x8": tail(tiles)}
        for name, (rh, rw) in {"sorted_in_32x8_quads": (8, 32), "sorted_in_32x32": (32, 32), "sorted_in_64x64": (64, 64), "sorted_in_128x8": (8, 128)}.items():
            hh, ww = h - h % rh, w - w % rw
            reg = cost[:hh, :ww].reshape(hh // rh, rh, ww // rw, rw).transpose(0, 2, 1, 3).reshape(-1, rh * rw)
            reg = np.sort(reg, axis=1)
            r[name] = tail(reg.reshape(-1, 64))
        r["sorted_globally"] = tail(np.sort(cost.reshape(-1)))
        res["spp_%d" % spp] = r
    # does the cost map of one launch predict another (different seeds)?  sort 64-spp pixels by the 8-spp map
    order = np.argsort(maps[8].reshape(h // 32, 32, w // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 1024), axis=1) if h % 32 == 0 else None
    hh, ww = h - h % 32, w - w % 32
    a8 = maps[8][:hh, :ww].reshape(hh // 32, 32, ww // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 1024)
    a64 = maps[64][:hh, :ww].reshape(hh // 32, 32, ww // 32, 32).transpose(0, 2, 1, 3).reshape(-1, 1024)
    idx = np.argsort(a8, axis=1, kind="stable")
    res["spp_64_sorted_in_32x32_by_the_8spp_map"] = tail(np.take_along_axis(a64, idx, 1).reshape(-1, 64))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
