#!/usr/bin/env python3
"""part_bound.py -- the ONE-GPU bound on the strong scaling of an image cut into row-stripe parts (BASELINE.json configs[3] and [4]):
times the whole image on one GPU, then EVERY one of the N parts as its rank would render it (same stripes, same global seeds, the
dispatch order recorded by nine warm-up launches), and predicts  speed-up(N GPUs) = whole / slowest part  before the gather.

    python tools/part_bound.py --config c4 > gpurun_out/c4_part.json     # C4: 3840x2160, 1024 spp, S16, render Inline
    python tools/part_bound.py --config c5 > gpurun_out/c5_part.json     # C5: 3840x2160, 512 spp, glass scene, render Streams, both forms

A part holds 1/N of the pixels but the same sample chain per pixel, so its launch has N x fewer waves of the same length and its end -- the
last waves on a half-empty chip -- weighs N x more than in the whole-image launch; and where the expensive pixels are (the glass spheres
do not spread over the rows the way S16's grid does) decides how evenly stripes of a given height deal them out.  Per stripe height the
table gives every part's time, the slowest, and the imbalance = slowest / mean.

C4 knows one more axis (`--chunks`): 0 = one launch with the in-kernel sample chunks (PTMI_OPT_SPP_CHUNKS automatic, the default of the
product), k = k chained launches with the chunks switched off.  C5 is timed as the product runs it (automatic everything), in both forms
of `render Streams`: the per-pixel tree walk (the default with GLASS) and the stream ("wavefront") form.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

CONFIGS = {
    "c4": dict(name="C4", scene="s16", algorithm="inline", spp=1024, forms=["auto"], chunks="0,1,8", stripes="8,10,6"),
    "c5": dict(name="C5", scene="glass", algorithm="streams", spp=512, forms=["tree", "stream"], chunks="0", stripes="8,10,6"),
}
LIMIT = 8


def timed(ctx, cam, spp, chunks, algorithm, pkg, inline):
    """One pass of `spp` samples; host clock around stream-ordered launches.  Inline: chunks == 0 is ONE launch with the in-kernel sample
    chunks, k > 0 is k chained launches with them switched off."""
    if inline:
        ctx.set_option(pkg.binding.OPT_SPP_CHUNKS, 0 if chunks == 0 else 1)
    n = max(chunks, 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render(cam, LIMIT, spp // n, algorithm)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e3


def measure(pkg, scene, cam, args, algorithm, form, chunk_list, partition=None, warm=9):
    """{chunks_K_ms: [repeats]} for one context: the whole image (partition None) or (stripe, parts, part)."""
    B = pkg.binding
    inline = algorithm == pkg.INLINE
    with pkg.Context(0) as c:
        c.set_scene(*scene)
        if partition:
            c.set_partition(*partition)
        c.resize(args.width, args.height)
        c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if form == "stream" else B.FORM_PIXEL)
        c.init_output(0x5EED1234)
        for _ in range(warm):                                  # the dispatch order is rebuilt before launch 1, 2, 4, 8: none inside the timed ones
            c.render(cam, LIMIT, args.spp, algorithm)
        rec = {"rows": c.local_rows}
        for ch in chunk_list:
            rec["chunks_%d_ms" % ch] = [round(timed(c, cam, args.spp, ch, algorithm, pkg, inline), 3) for _ in range(args.repeats)]
        if not inline:
            c.reset_stats()
            c.render(cam, LIMIT, args.spp, algorithm)
            st = c.stats()
            rec["rays_spilled_overflowed_dropped_truncated"] = [st["stream_rays_spilled"], st["stream_rays_overflowed"], st["stream_rays_dropped"], st["stream_rays_truncated"]]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c4")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--stripes", default=None)
    ap.add_argument("--chunks", default=None, help="Inline only: 0 = one launch with in-kernel sample chunks, k = k chained launches without")
    ap.add_argument("--forms", default=None, help="render Streams only: tree,stream")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--warm", type=int, default=9)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.spp = args.spp or cfg["spp"]
    pkg = graft.load_package()
    pkg._build.build_lib()
    scene = {"s16": pkg.world.scene16, "glass": pkg.world.glass_scene}[cfg["scene"]]()
    algorithm = pkg.INLINE if cfg["algorithm"] == "inline" else pkg.STREAMS
    cam = pkg.world.initial_camera()
    chunk_list = [int(c) for c in (args.chunks or cfg["chunks"]).split(",")]
    forms = (args.forms.split(",") if args.forms else cfg["forms"])
    stripes = [int(s) for s in (args.stripes or cfg["stripes"]).split(",")]
    out = {"workload": "%s: %dx%d, %d spp, limit %d, scene %s, render %s, %d row-stripe parts" % (
               cfg["name"], args.width, args.height, args.spp, LIMIT, cfg["scene"].upper(), cfg["algorithm"].capitalize(), args.parts),
           "binary_build_id": pkg.load_library().build_id, "forms": {}}

    for form in forms:
        whole = measure(pkg, scene, cam, args, algorithm, form, chunk_list, None, args.warm)
        whole_ms = min(min(v) for k, v in whole.items() if k.endswith("_ms"))
        whole["best_ms"] = whole_ms
        print("%s, whole image: %.2f ms" % (form, whole_ms), file=sys.stderr, flush=True)
        res = {"whole": whole, "stripes": {}}
        best = None
        for stripe in stripes:
            rows = []
            for part in range(args.parts):
                rec = measure(pkg, scene, cam, args, algorithm, form, chunk_list, (stripe, args.parts, part), args.warm)
                rec["part"] = part
                rows.append(rec)
            summary = {"parts": rows}
            for ch in chunk_list:
                k = "chunks_%d_ms" % ch
                per_part = [min(r[k]) for r in rows]
                slowest = max(per_part)
                summary["per_part_%s" % k] = per_part
                summary["slowest_part_%s" % k] = slowest
                summary["imbalance_%s" % k.replace("_ms", "")] = round(slowest / (sum(per_part) / len(per_part)), 4)
                summary["predicted_speedup_%d_gpus_%s" % (args.parts, k.replace("_ms", ""))] = round(whole_ms / slowest, 3)
                if best is None or slowest < best[0]:
                    best = (slowest, stripe, ch)
            summary["ideal_part_ms"] = round(whole_ms / args.parts, 3)
            res["stripes"][str(stripe)] = summary
            print("%s, stripe %d: %s" % (form, stripe, {k: v for k, v in summary.items() if k != "parts"}), file=sys.stderr, flush=True)
        res["best"] = {"slowest_part_ms": best[0], "stripe_rows": best[1], "chunks": best[2],
                       "predicted_speedup": round(whole_ms / best[0], 3),
                       "predicted_efficiency": round(whole_ms / best[0] / args.parts, 4),
                       "per_part_over_whole": round(best[0] / whole_ms, 4)}
        out["forms"][form] = res
    if args.config == "c4":                                   # the layout profiles/r0N_c4_part.json has had since round 2 (tools/design_numbers.py reads it)
        only = out["forms"]["auto"]
        out.update({"whole": only["whole"], "stripes": only["stripes"], "best": only["best"]})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
