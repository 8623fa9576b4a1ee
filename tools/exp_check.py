#!/usr/bin/env python3
"""exp_check.py LIB.so [...] -- do experimental builds of libptmi (tools/ab.py build NAME -DFLAG) still render the product's planes?
C2's shape at 8 spp (render Inline, S16, limit 8) and mainScene at limit 15, all seven planes against the default library's, bit for bit
(the default library is what the parity suite holds against the oracle)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
pkg._build.build_lib()
cam = pkg.world.initial_camera()
CASES = [("s16", pkg.world.scene16(), 1920, 1080, 8, 8), ("main", pkg.world.main_scene(), 800, 600, 15, 5)]


def render(lib):
    out = []
    for _name, (sp, pl), w, h, limit, spp in CASES:
        with pkg.Context(0, library=lib) as c:
            c.set_scene(sp, pl)
            c.resize(w, h)
            c.init_output(0x5EED1234)
            c.render(cam, limit, spp)
            out.append(c.download_state())
    return out


want = render(pkg.load_library())
for path in sys.argv[1:]:
    got = render(pkg.binding.open_library(path, check_build_id=False))
    bad = [(case[0], k) for case, a, b in zip(CASES, got, want) for k in range(7) if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))]
    print("%-40s %s" % (path.rsplit("/", 1)[-1], "bit-identical to the default library" if not bad else "DIFFERS: %s" % bad))
