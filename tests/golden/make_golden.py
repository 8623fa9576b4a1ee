#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

PROVENANCE: "self-oracle".  The reference (Haskell + Accelerate + LLVM 9) cannot be built or run in
this environment and holds no golden images or vectors for the render path, so these planes come
from the repo's own CPU oracle (oracle/pt_oracle.c) at the commit that introduced them.  They pin
the oracle (and through it the HIP path) against silent drift; they do NOT pin it against GHC.
The intersection path IS pinned against the reference's own tests (tests/refprops.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402

CASES = {
    # name: (scene, width, height, bounce_limit, n_spp, algorithm)
    "main_64x48_l4_s1": ("main", 64, 48, 4, 1, "inline"),
    "main_64x48_l15_s2": ("main", 64, 48, 15, 2, "inline"),
    "s16_80x45_l8_s4": ("s16", 80, 45, 8, 4, "inline"),
    # render Streams under the library's default seed rule (PTMI_SEED_AUTO = `combine new old` without GLASS) ...
    "main_64x48_streams_s2": ("main", 64, 48, 1 << 16, 2, "streams"),
    # ... and under the other reading of `combine` (assumption A5): the pixel keeps the accumulator's seed
    "main_64x48_streams_keep_s2": ("main", 64, 48, 1 << 16, 2, "streams_keep"),
    # build-defined GLASS extension (no reference semantics at all): stream form, hard cap 64 steps
    "glass_64x48_wavefront_s2": ("glass", 64, 48, 64, 2, "wavefront"),
}
SEED0 = 0x5EED1234


def main():
    pkg, ora = graft.load_package(), graft.load_oracle()
    cam = pkg.world.initial_camera()
    for name, (scene, w, h, limit, spp, alg) in CASES.items():
        sp, pl = {"main": pkg.world.main_scene, "s16": pkg.world.scene16, "glass": pkg.world.glass_scene}[scene]()
        seeds = ora.gen_seeds(SEED0, 0, w * h)
        start = [np.zeros((h, w), np.float32)] * 3 + [s.reshape(h, w) for s in seeds]
        if alg == "inline":
            out, live = ora.render_inline(sp, pl, cam, w, h, limit, spp, start)
        elif alg == "streams":
            assert ora.default_seed_rule(sp, pl) == ora.SEED_FROM_RESULT
            out, live = ora.render_streams(sp, pl, cam, w, h, limit, spp, start)
        elif alg == "streams_keep":
            out, live = ora.render_streams(sp, pl, cam, w, h, limit, spp, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)
        else:
            out, live, dropped, _steps = ora.render_streams_wavefront(sp, pl, cam, w, h, limit, spp, start)
            assert dropped == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), scene=scene, width=w, height=h, limit=limit,
                            spp=spp, algorithm=alg, seed0=SEED0, live=live,
                            spheres=sp, planes=pl, camera=cam,
                            **{"in_%d" % i: a for i, a in enumerate(start)},
                            **{"out_%d" % i: a for i, a in enumerate(out)})
        print(name, "live bounces", live)


if __name__ == "__main__":
    main()
