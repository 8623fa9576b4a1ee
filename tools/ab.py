#!/usr/bin/env python3
"""ab.py -- A/B timing of differently-flagged builds of libptmi on the standard workloads.

    python tools/ab.py build NAME -DFLAG ...      # in the container: hipcc -> build/ab/NAME.so (travels with gpurun)
    python tools/ab.py run [--workloads c2,glass_tree] [build/ab/*.so]    # on the GPU box: one subprocess per library

Workloads (1080p, 64 spp, ms per launch, best of N after warm-up): c2 (render Inline, scene S16), streams (per-pixel
Streams, S16), s16_stream (stream form, S16), glass_tree (per-pixel tree walk, glass scene), glass_stream (stream form); c5_tree /
c5_stream: one of 8 parts of the 4K / 512-spp glass image; suffixes _uniform (PTMI_OPT_STREAM_GRADED = 0), _gK (PTMI_OPT_GLASS_BATCH = K),
_bK (PTMI_OPT_STREAM_BATCH = K)."""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

# name -> (scene, algorithm, form, options {PTMI_OPT_* name without the prefix: value}, shape or None = 1080p / 64 spp whole image)
C5_PART = (3840, 2160, 512, 8)      # one of 8 parts (10-row stripes) of BASELINE configs[4]
WORKLOADS = {
    "c2": ("s16", "inline", "auto", {}, None), "streams": ("s16", "streams", "auto", {}, None), "s16_stream": ("s16", "streams", "stream", {}, None),
    "glass_tree": ("glass", "streams", "auto", {}, None), "glass_stream": ("glass", "streams", "stream", {}, None),
    "glass_stream_uniform": ("glass", "streams", "stream", {"STREAM_GRADED": 0}, None),
    "c5_tree": ("glass", "streams", "auto", {}, C5_PART), "c5_stream": ("glass", "streams", "stream", {}, C5_PART),
    "c5_stream_uniform": ("glass", "streams", "stream", {"STREAM_GRADED": 0}, C5_PART),
}
for _k in (2, 4, 8, 16, 32):
    WORKLOADS["glass_stream_g%d" % _k] = ("glass", "streams", "stream", {"GLASS_BATCH": _k}, None)
    WORKLOADS["c5_stream_g%d" % _k] = ("glass", "streams", "stream", {"GLASS_BATCH": _k}, C5_PART)
for _k in (1, 2, 4, 6, 8, 12, 16):                          # PTMI_OPT_SPP_CHUNKS of the per-tile kernels (0 = automatic is the plain workload)
    WORKLOADS["glass_tree_c%d" % _k] = ("glass", "streams", "auto", {"SPP_CHUNKS": _k}, None)
    WORKLOADS["c5_tree_c%d" % _k] = ("glass", "streams", "auto", {"SPP_CHUNKS": _k}, C5_PART)
for _k in (0, 50, 100, 150, 200, 250, 300, 400, 500):      # PTMI_OPT_STREAM_TAIL (thousandths of the recorded cost left to the per-pixel kernel)
    WORKLOADS["s16_stream_t%d" % _k] = ("s16", "streams", "stream", {"STREAM_TAIL": _k}, None)
WORKLOADS["c2_v17"] = ("s16", "inline", "auto", {"VARIANT": 17}, None)     # render Inline with the scene through scalar loads instead of LDS (ptmi_set_variant 17)
WORKLOADS["c2_main"] = ("main", "inline", "auto", {}, None)                # ... on mainScene (7 primitives)
C4_PART = (3840, 2160, 1024, 8)     # one of 8 parts of BASELINE configs[3]
C4_PART_256 = (3840, 2160, 256, 8)
for _tag, _shape in (("c4p", C4_PART), ("c4p256", C4_PART_256)):           # the stream form on one part, S16: ordered passes and their hand-off
    WORKLOADS[_tag + "_stream_1pass"] = ("s16", "streams", "stream", {"ORDERED_PASSES": 1}, _shape)
    WORKLOADS[_tag + "_stream_auto"] = ("s16", "streams", "stream", {}, _shape)
    for _k in (4, 8):
        WORKLOADS[_tag + "_stream_%dfenced" % _k] = ("s16", "streams", "stream", {"ORDERED_PASSES": _k, "PASS_HANDOFF": 0}, _shape)
        WORKLOADS[_tag + "_stream_%dfree" % _k] = ("s16", "streams", "stream", {"ORDERED_PASSES": _k, "PASS_HANDOFF": 1}, _shape)
WORKLOADS["c4_part"] = ("s16", "inline", "auto", {}, C4_PART)
WORKLOADS["c4_part_streams"] = ("s16", "streams", "auto", {}, C4_PART)
for _k in (1, 2, 4, 5, 6, 8, 12, 16):
    WORKLOADS["c4_part_c%d" % _k] = ("s16", "inline", "auto", {"SPP_CHUNKS": _k}, C4_PART)
    WORKLOADS["c4_part_streams_c%d" % _k] = ("s16", "streams", "auto", {"SPP_CHUNKS": _k}, C4_PART)
for _k in (102, 103, 104, 106, 107, 113, 164):                               # ... groups of k - 100 passes all the way
    WORKLOADS["glass_stream_s%d" % _k] = ("glass", "streams", "stream", {"STREAM_PASS_GROUPS": _k}, None)
    WORKLOADS["c5_stream_s%d" % _k] = ("glass", "streams", "stream", {"STREAM_PASS_GROUPS": _k}, C5_PART)
for _k in (1, 2, 3, 4, 5, 6, 7, 8):                           # PTMI_OPT_STREAM_PASS_GROUPS: the last k passes region by region
    WORKLOADS["glass_stream_s%d" % _k] = ("glass", "streams", "stream", {"STREAM_PASS_GROUPS": _k}, None)
    WORKLOADS["c5_stream_s%d" % _k] = ("glass", "streams", "stream", {"STREAM_PASS_GROUPS": _k}, C5_PART)
for _k in (4, 8, 16, 32, 64):
    WORKLOADS["glass_stream_b%d" % _k] = ("glass", "streams", "stream", {"STREAM_BATCH": _k}, None)
    WORKLOADS["s16_stream_b%d" % _k] = ("s16", "streams", "stream", {"STREAM_BATCH": _k}, None)


def one(lib, names, width=1920, height=1080, spp=64, repeats=7):
    if os.environ.get("PTMI_AB_SHAPE"):                    # e.g. 3840x2160x64
        width, height, spp = (int(v) for v in os.environ["PTMI_AB_SHAPE"].split("x"))
    pkg = graft.load_package()
    if lib != "default":
        pkg.binding._lib = None
        pkg.binding.load_library(lib, check_build_id=False)   # an A/B partner is built from OTHER sources on purpose; its id is printed
    else:
        pkg._build.build_lib()
    B = pkg.binding
    cam = pkg.world.initial_camera()
    out = {"build_id": pkg.load_library().build_id}
    # (the first workload a process times is ~0.5 % slower than the same workload later -- clocks, cold allocations: it is run once unrecorded)
    for k, name in enumerate([names[0]] + list(names)):
        scene, alg, form, options, shape = WORKLOADS[name]
        w, h, n_spp, parts = shape if shape else (width, height, spp, 1)
        sp, pl = {"s16": pkg.world.scene16, "glass": pkg.world.glass_scene, "main": pkg.world.main_scene}[scene]()
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            if parts > 1:
                c.set_partition(10, parts, 0)
            c.resize(w, h)
            c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if form == "stream" else B.FORM_PIXEL)   # ("auto" workloads are the per-pixel kernels, also on a part)
            for opt, value in options.items():
                if opt == "VARIANT":
                    c.set_variant(value)
                else:
                    c.set_option(getattr(B, "OPT_" + opt), value)
            c.init_output(0x5EED1234)
            algorithm = pkg.INLINE if alg == "inline" else pkg.STREAMS
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:                       # clock ramp + cost order
                c.render(cam, 8, n_spp, algorithm)
                c.synchronize()
            times = []
            for _ in range(repeats):
                c.synchronize()
                t0 = time.perf_counter()
                c.render(cam, 8, n_spp, algorithm)
                c.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
            if k == 0:
                continue
            out[name] = round(min(times), 3)
            if form == "stream":
                c.reset_stats()
                c.render(cam, 8, n_spp, algorithm)
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
