#!/usr/bin/env python3
"""soak_passes.py [launches] [fenced|fence_free] -- the ordered passes of the stream form under load: C2's image (1920x1080, 64 spp, S16) rendered by
the per-pixel chain kernel and, with the same seeds, by the stream form: in turn as whole sample chains (one pass, the cheap end of
the dispatch order rendered by the chain kernel beside the persistent launch) and with the chains cut into 64, 32, 16, 8 and 4 ordered
passes (items of 1, 2, 4, 8, 16 samples: up to 130 million hand-offs per launch, lane to lane through the planes -- by default with the
library's default hand-off, release / acquire once per region and pass; `fence_free`: write-through stores + counter + sc1 loads), while a
third context renders render Inline on a stream of its own beside them.
Every launch: all seven planes of the two forms compared bit for bit.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    handoff = sys.argv[2] if len(sys.argv) > 2 else "fenced"
    pkg = graft.load_package()
    pkg._build.build_lib()
    B = pkg.binding
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    handoffs = 0
    with pkg.Context(0) as chain, pkg.Context(0) as stream, pkg.Context(0) as noise:
        for c in (chain, stream, noise):
            c.set_scene(sp, pl)
            c.resize(1920, 1080)
            c.init_output(0x5EED1234)
        stream.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        stream.set_option(B.OPT_PASS_HANDOFF, B.HANDOFF_FENCE_FREE if handoff == "fence_free" else B.HANDOFF_FENCED)
        hits = None
        for k in range(launches):
            batch = (0, 1, 2, 4, 8, 16)[k % 6]              # 0: whole sample chains, with the per-pixel tail beside the persistent launch
            stream.set_option(B.OPT_STREAM_BATCH, batch)
            noise.render(cam, 8, 64, pkg.INLINE)                  # asynchronous: runs beside the two below
            chain.render(cam, 8, 64, pkg.STREAMS)
            stream.render(cam, 8, 64, pkg.STREAMS)
            a, b = stream.download_state(), chain.download_state()
            for i, (x, y) in enumerate(zip(a, b)):
                if not np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)):
                    bad = int(np.count_nonzero(np.asarray(x).view(np.uint32) != np.asarray(y).view(np.uint32)))
                    print(json.dumps({"ok": False, "launch": k, "samples_per_item": batch, "plane": i, "words_differing": bad}))
                    sys.exit(1)
            if hits is None:
                hits = int(np.count_nonzero(np.asarray(a[0]) != 0.0))   # a lower bound of the pixels with a start hit: those whose colour is not zero after one launch
            handoffs += hits * (64 // batch - 1) if batch else 0
            if k % 10 == 9:
                print("launch %d ok" % (k + 1), file=sys.stderr, flush=True)
    print(json.dumps({"ok": True, "launches": launches, "image": "1920x1080, 64 spp, S16", "samples_per_item_cycle": ["all (one pass, per-pixel tail)", 1, 2, 4, 8, 16],
                      "handoffs_between_lanes_at_least": handoffs, "compared": "all seven planes, bit for bit, against the per-pixel chain kernel, every launch",
                      "beside": "a third context rendering render Inline on its own stream", "handoff": handoff,
                      "binary_build_id": pkg.load_library().build_id}))


if __name__ == "__main__":
    main()
