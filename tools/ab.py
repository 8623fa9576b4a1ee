#!/usr/bin/env python3
"""ab.py -- A/B timing of differently-flagged builds of libptmi on the standard workloads.

    python tools/ab.py build NAME -DFLAG ...      # in the container: hipcc -> build/ab/NAME.so (travels with gpurun)
    python tools/ab.py run [--workloads c2,glass_tree] [build/ab/*.so]    # on the GPU box: one subprocess per library

Workloads (1080p, 64 spp, ms per launch, best of N after warm-up): c2 (render Inline, scene S16), streams (per-pixel
Streams, S16), s16_stream (stream form, S16), glass_tree (per-pixel tree walk, glass scene), glass_stream (stream form)."""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

WORKLOADS = {
    "c2": ("s16", "inline", "auto"), "streams": ("s16", "streams", "auto"), "s16_stream": ("s16", "streams", "stream"),
    "glass_tree": ("glass", "streams", "auto"), "glass_stream": ("glass", "streams", "stream"),
    "s16_stream_b16": ("s16", "streams", "stream16"), "glass_stream_b8": ("glass", "streams", "stream8"),
    "glass_stream_b32": ("glass", "streams", "stream32"), "glass_stream_b16": ("glass", "streams", "stream16"),
    "glass_stream_b64": ("glass", "streams", "stream64"), "glass_stream_b4": ("glass", "streams", "stream4"),
}


def one(lib, names, width=1920, height=1080, spp=64, repeats=7):
    if os.environ.get("PTMI_AB_SHAPE"):                    # e.g. 3840x2160x64
        width, height, spp = (int(v) for v in os.environ["PTMI_AB_SHAPE"].split("x"))
    pkg = graft.load_package()
    if lib != "default":
        pkg.binding._lib = None
        pkg.binding.load_library(lib)
    else:
        pkg._build.build_lib()
    cam = pkg.world.initial_camera()
    out = {}
    for name in names:
        scene, alg, form = WORKLOADS[name]
        sp, pl = {"s16": pkg.world.scene16, "glass": pkg.world.glass_scene}[scene]()
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.resize(width, height)
            c.set_option(pkg.binding.OPT_STREAMS_FORM, pkg.binding.FORM_STREAM if form.startswith("stream") else pkg.binding.FORM_AUTO)
            if form.startswith("stream") and form[6:]:
                c.set_option(pkg.binding.OPT_STREAM_BATCH, int(form[6:]))
            c.init_output(0x5EED1234)
            algorithm = pkg.INLINE if alg == "inline" else pkg.STREAMS
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:                       # clock ramp + cost order
                c.render(cam, 8, spp, algorithm)
                c.synchronize()
            times = []
            for _ in range(repeats):
                c.synchronize()
                t0 = time.perf_counter()
                c.render(cam, 8, spp, algorithm)
                c.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
            out[name] = round(min(times), 3)
            if form.startswith("stream"):
                c.reset_stats()
                c.render(cam, 8, spp, algorithm)
                st = c.stats()
                if st["stream_rays_spilled"] or st["stream_rays_dropped"]:
                    out[name + "_spilled/overflowed/dropped"] = [st["stream_rays_spilled"], st["stream_rays_overflowed"], st["stream_rays_dropped"]]
    return out


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "build":
        pkg = graft.load_package()
        os.makedirs(os.path.join(ROOT, "build", "ab"), exist_ok=True)
        out = os.path.join(ROOT, "build", "ab", sys.argv[2] + ".so")
        pkg._build.build_lib(out=out, extra_flags=sys.argv[3:])
        print(out)
        return
    if len(sys.argv) >= 2 and sys.argv[1] == "one":
        print(json.dumps(one(sys.argv[2], sys.argv[3].split(","))))
        return
    args = sys.argv[2:] if len(sys.argv) >= 2 and sys.argv[1] == "run" else sys.argv[1:]
    names = ["c2", "streams", "s16_stream", "glass_tree", "glass_stream"]
    if args and args[0] == "--workloads":
        names = args[1].split(",")
        args = args[2:]
    libs = ["default"] + (args if args else sorted(glob.glob(os.path.join(ROOT, "build", "ab", "*.so"))))
    for lib in libs:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "one", lib, ",".join(names)], capture_output=True, text=True)
        tag = os.path.basename(lib)
        print("%-28s %s" % (tag, res.stdout.strip() if res.returncode == 0 else "FAILED: " + res.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
