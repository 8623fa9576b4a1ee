#!/usr/bin/env python3
"""c4_part.py -- bounds the strong-scaling drain of BASELINE.json configs[3] (C4: ONE 3840x2160 image at 1024 spp,
row-striped over 8 GPUs) on ONE GPU: times the whole image, then every one of the 8 parts as its rank would render
it (same stripes, same global seeds), as one 1024-spp launch and as chained sub-launches (8 x 128 spp, ...).

A part holds 1/8 of the pixels but the same 1024-sample chain per pixel, so the launch has 8x fewer waves of the
same length and its end -- the last waves on a half-empty chip -- weighs 8x more than in the whole-image launch.
predicted_speedup_8 = whole-image time / slowest part (the gather of 12.4 MB per rank overlaps the next render).

    python tools/c4_part.py [--spp 1024] [--stripes 8,10,6] > gpurun_out/c4_part.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def timed(ctx, cam, limit, spp, chunks, pkg):
    """One pass of `spp` samples as `chunks` equal launches; host clock around stream-ordered launches.
    chunks == 0: ONE launch with the in-kernel sample chunks (PTMI_OPT_SPP_CHUNKS automatic); every other
    figure is taken with them switched off."""
    ctx.set_option(pkg.binding.OPT_SPP_CHUNKS, 0 if chunks == 0 else 1)
    chunks = max(chunks, 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(chunks):
        ctx.render(cam, limit, spp // chunks, pkg.INLINE)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--stripes", default="8,10,6")
    ap.add_argument("--chunks", default="0,1,8", help="0 = one launch with in-kernel sample chunks, k = k chained launches without")
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    pkg = graft.load_package()
    pkg._build.build_lib()
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    limit = 8
    chunk_list = [int(c) for c in args.chunks.split(",")]
    out = {"workload": "C4: %dx%d, %d spp, limit %d, scene S16, %d row-stripe parts" % (args.width, args.height, args.spp, limit, args.parts),
           "whole": {}, "stripes": {}}

    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        c.resize(args.width, args.height)
        c.init_output(0x5EED1234)
        for ch in chunk_list:
            out["whole"]["chunks_%d_ms" % ch] = [round(timed(c, cam, limit, args.spp, ch, pkg), 3) for _ in range(args.repeats)]
    whole_ms = min(min(v) for v in out["whole"].values())
    out["whole"]["best_ms"] = whole_ms
    print("whole image: %.2f ms" % whole_ms, file=sys.stderr, flush=True)

    best = None
    for stripe in [int(s) for s in args.stripes.split(",")]:
        rows = []
        for part in range(args.parts):
            with pkg.Context(0) as c:
                c.set_scene(sp, pl)
                c.set_partition(stripe, args.parts, part)
                c.resize(args.width, args.height)
                c.init_output(0x5EED1234)
                rec = {"part": part, "rows": c.local_rows}
                for ch in chunk_list:
                    rec["chunks_%d_ms" % ch] = [round(timed(c, cam, limit, args.spp, ch, pkg), 3) for _ in range(args.repeats)]
                rows.append(rec)
        summary = {"parts": rows}
        for ch in chunk_list:
            k = "chunks_%d_ms" % ch
            slowest = max(min(r[k]) for r in rows)
            summary["slowest_part_%s" % k] = slowest
            summary["predicted_speedup_%d_gpus_%s" % (args.parts, k.replace("_ms", ""))] = round(whole_ms / slowest, 3)
            if best is None or slowest < best[0]:
                best = (slowest, stripe, ch)
        summary["ideal_part_ms"] = round(whole_ms / args.parts, 3)
        out["stripes"][str(stripe)] = summary
        print("stripe %d: %s" % (stripe, {k: v for k, v in summary.items() if k != "parts"}), file=sys.stderr, flush=True)
    out["best"] = {"slowest_part_ms": best[0], "stripe_rows": best[1], "chunks": best[2],
                   "predicted_speedup": round(whole_ms / best[0], 3),
                   "predicted_efficiency": round(whole_ms / best[0] / args.parts, 4),
                   "per_part_over_whole": round(best[0] / whole_ms, 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
