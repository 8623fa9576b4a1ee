#!/usr/bin/env python3
"""drain_fit.py -- how much of a launch is the end-of-kernel drain?  Times the default kernel (variant 0: tiles
dispatched most expensive first) and variant 13 (image order) on images of 1920 x {270 ... 4320} pixels, 64 spp, and
fits time = a + b * pixels: b is the steady-state cost per pixel, a what a launch pays for starting and draining."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    pkg._build.build_lib()
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    out = {}
    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        for variant in (0, 13):
            rows = []
            for h in (270, 540, 1080, 2160, 4320):
                c.resize(1920, h)
                c.init_output(0x5EED1234)
                c.set_variant(variant)
                for _ in range(6):
                    c.render(cam, 8, 64)
                c.synchronize()
                n = 10
                t0 = time.perf_counter()
                for _ in range(n):
                    c.render(cam, 8, 64)
                c.synchronize()
                rows.append((1920 * h, (time.perf_counter() - t0) / n * 1e3))
            x = np.array([r[0] for r in rows], float)
            y = np.array([r[1] for r in rows], float)
            b, a = np.polyfit(x, y, 1)
            out["variant_%d" % variant] = {"ms_by_pixels": {int(p): round(t, 4) for p, t in rows},
                                           "a_ms": round(float(a), 4), "b_ns_per_pixel": round(float(b) * 1e6, 4),
                                           "drain_share_at_1080p": round(float(a) / rows[2][1], 4)}
            print(json.dumps({("variant_%d" % variant): out["variant_%d" % variant]}), flush=True)


if __name__ == "__main__":
    main()
