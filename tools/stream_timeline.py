#!/usr/bin/env python3
"""stream_timeline.py -- run under `rocprofv3 --kernel-trace`: a few stream-form renders of S16 at 1080p / 64 spp, so that
the trace shows how the launches of one render call overlap (tools/show_timeline.py prints the last call's kernels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
pkg._build.build_lib()
B = pkg.binding
scene = sys.argv[1] if len(sys.argv) > 1 else "s16"
sp, pl = pkg.world.scene16() if scene == "s16" else pkg.world.glass_scene()
with pkg.Context(0) as c:
    c.set_scene(sp, pl)
    c.resize(1920, 1080)
    c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
    c.init_output(0x5EED1234)
    for _ in range(40):
        c.render(pkg.world.initial_camera(), 8, 64, pkg.STREAMS)
        c.synchronize()
