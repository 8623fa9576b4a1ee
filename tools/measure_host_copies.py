#!/usr/bin/env python3
"""Host-buffer entry points (PCIe-inclusive; never the headline): ptmi_render1 (7 planes in, 7 out, one sample),
ptmi_download_color and ptmi_present at 800x600 (reference-native), 1080p and 4K, with the staging engine off
(PTMI_STAGE_THREADS=0: plain hipMemcpyAsync from pageable memory) and on (default and 16 threads).
Output arrays are allocated fresh per call, as the Haskell caller's are."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def best_of(fn, n=7):
    fn(); fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return round(ts[len(ts) // 2] * 1e3, 3)


def main():
    pkg = graft.load_package()
    pkg._build.build_lib()
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    out = {"host_threads": os.cpu_count()}
    for threads in ("0", None, "16"):
        if threads is None:
            os.environ.pop("PTMI_STAGE_THREADS", None)
        else:
            os.environ["PTMI_STAGE_THREADS"] = threads
        row = {}
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            for name, (w, h) in (("800x600", (800, 600)), ("1080p", (1920, 1080)), ("4k", (3840, 2160))):
                ctx.resize(w, h)
                ctx.init_output(0x5EED1234)
                state = {"planes": ctx.download_state()}

                def r1():
                    state["planes"] = ctx.render1(cam, 8, w, h, state["planes"])
                row["render1_%s_ms" % name] = best_of(r1)
                row["render1_%s_GBps_both_ways" % name] = round(2 * 28 * w * h / (row["render1_%s_ms" % name] * 1e-3) / 1e9, 1)
                row["download_color_%s_ms" % name] = best_of(ctx.download_color)
                row["present_rgba8_%s_ms" % name] = best_of(lambda: ctx.present(4, rgb32f=False, rgba8=True))
                row["upload_state_%s_ms" % name] = best_of(lambda: ctx.upload_state(*state["planes"]))
        out["engine_off" if threads == "0" else ("engine_default" if threads is None else "engine_16_threads")] = row
        print(json.dumps({threads or "default": row}), flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "host_copies.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
