#!/usr/bin/env python3
"""c0_calls.py [lib.so] -- the reference's own configuration (800x600, mainScene, limit 15): microseconds per resident
ptmi_render call of 1, 2, 4, 8 and 30 samples (`compileFor`'s closure renders one sample per call; computationLoop batches
at least 30, app/Main.hs:209-211)."""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
if len(sys.argv) > 1:
    pkg.binding._lib = None
    pkg.binding.load_library(sys.argv[1])
cam = pkg.world.initial_camera()
out = {}
with pkg.Context(0) as ctx:
    ctx.set_scene(*pkg.world.main_scene())
    ctx.resize(800, 600)
    ctx.init_output(1)
    for n in (1, 2, 4, 8, 30):
        for _ in range(50):
            ctx.render(cam, 15, n)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.render(cam, 15, n)
        ctx.synchronize()
        out["%d_spp_us" % n] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
print(json.dumps(out))
