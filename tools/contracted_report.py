#!/usr/bin/env python3
"""contracted_report.py -- what the literal reading of the reference's arithmetic costs, and how far a contracting compiler
would move the picture (PTMI_OPT_ARITHMETIC = PTMI_ARITH_CONTRACTED: the same Inline kernel with a * b + c fused; a labelled
measurement mode, never the default and never the headline).  For C0 (800x600, mainScene, limit 15) and C2 (1920x1080, S16,
limit 8): ms per 64-spp launch in both modes, and after 1, 8 and 64 samples from the same seeds the share of pixels whose
three colour sums are within 1e-4 relative of the exact mode's (= the oracle's, bit for bit), the share that is bit-identical,
and the share whose RNG state still equals the exact mode's (a path that takes a different branch draws differently)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    pkg._build.build_lib()
    B = pkg.binding
    cam = pkg.world.initial_camera()
    out = {}
    for name, scene, w, h, limit in (("C0", pkg.world.main_scene(), 800, 600, 15), ("C2", pkg.world.scene16(), 1920, 1080, 8)):
        rec = {}
        with pkg.Context(0) as c:
            c.set_scene(*scene)
            c.resize(w, h)
            for mode, tag in ((B.ARITH_EXACT, "exact"), (B.ARITH_CONTRACTED, "contracted")):
                c.set_option(B.OPT_ARITHMETIC, mode)
                c.init_output(0x5EED1234)
                t_end = time.perf_counter() + 0.25
                while time.perf_counter() < t_end:
                    c.render(cam, limit, 64)
                    c.synchronize()
                times = []
                for _ in range(7):
                    c.synchronize()
                    t0 = time.perf_counter()
                    c.render(cam, limit, 64)
                    c.synchronize()
                    times.append((time.perf_counter() - t0) * 1e3)
                rec["ms_per_64spp_" + tag] = round(min(times), 4)
            states = {}
            for mode, tag in ((B.ARITH_EXACT, "exact"), (B.ARITH_CONTRACTED, "contracted")):
                c.set_option(B.OPT_ARITHMETIC, mode)
                c.init_output(0x5EED1234)
                done = 0
                for upto in (1, 8, 64):
                    c.render(cam, limit, upto - done)
                    done = upto
                    states[(tag, upto)] = c.download_state()
            c.set_option(B.OPT_ARITHMETIC, B.ARITH_EXACT)
        for upto in (1, 8, 64):
            a, b = states[("contracted", upto)], states[("exact", upto)]
            close = np.ones((h, w), bool)
            same = np.ones((h, w), bool)
            for k in range(3):
                close &= np.abs(a[k] - b[k]) <= 1e-4 * np.abs(b[k])
                same &= a[k].view(np.uint32) == b[k].view(np.uint32)
            rng = np.ones((h, w), bool)
            for k in range(3, 7):
                rng &= a[k] == b[k]
            rec["after_%d_spp" % upto] = {"pixels_within_1e-4": round(float(close.mean()), 6), "pixels_bit_identical": round(float(same.mean()), 6),
                                          "pixels_with_the_same_rng_state": round(float(rng.mean()), 6)}
        rec["speedup"] = round(rec["ms_per_64spp_exact"] / rec["ms_per_64spp_contracted"], 4)
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
