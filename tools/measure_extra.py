#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the headline bench): the PCIe-inclusive rate of the
compat entry ptmi_render1 (host planes in/out, one sample), the kernel variants on C2, and C3."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402



def main():
    pkg = graft.load_package()
    pkg._build.build_lib()
    pkg.binding.load_library(pkg._build.build_ablations_lib())      # the variant table needs the ablation kernels (-DPTMI_ABLATIONS)
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    out = {}
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        # --- compat entry, 1080p, 1 sample, 8 bounces: H2D 58 MB + kernel + D2H 58 MB
        w, h = 1920, 1080
        ctx.resize(w, h)
        ctx.init_output(0x5EED1234)
        planes = ctx.download_state()
        for _ in range(2):
            planes = ctx.render1(cam, 8, w, h, planes)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            planes = ctx.render1(cam, 8, w, h, planes)
        dt = (time.perf_counter() - t0) / n
        out["render1_1080p_ms"] = round(dt * 1e3, 3)
        out["render1_1080p_Msamples_s"] = round(w * h * 8 / dt / 1e6, 1)
        out["render1_bytes_moved_MB"] = round(2 * 7 * w * h * 4 / 1e6, 1)
        # --- variants on C2 (resident), kernel time by host clock around synchronize
        res = {}
        t_ramp = time.perf_counter()                      # leave the idle clock state first (as bench.py does)
        while time.perf_counter() - t_ramp < 0.5:
            ctx.render(cam, 8, 64); ctx.synchronize()
        for v in (0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18):
            ctx.set_variant(v)
            ctx.init_output(0x5EED1234)
            for _ in range(3):
                ctx.render(cam, 8, 64)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.render(cam, 8, 64)
            ctx.synchronize()
            res[v] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        out["c2_ms_by_variant"] = res
        # --- Streams on C2: per-pixel form, stream form (ordered, and 16 samples per stream), both seed rules
        B = pkg.binding
        ctx.set_variant(0)

        def streams_ms(n=5):
            ctx.init_output(0x5EED1234)
            for _ in range(2):
                ctx.render(cam, 8, 64, pkg.STREAMS)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ctx.render(cam, 8, 64, pkg.STREAMS)
            ctx.synchronize()
            return round((time.perf_counter() - t0) / n * 1e3, 3)

        out["c2_streams_ms"] = streams_ms()                                  # the default seed rule: the result's seed (PTMI_SEED_AUTO)
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_KEEP_ACCUMULATOR)
        out["c2_streams_keep_accumulator_ms"] = streams_ms()
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        out["c2_streams_stream_form_keep_accumulator_ms"] = streams_ms(3)
        ctx.set_option(B.OPT_STREAM_BATCH, 16)
        out["c2_streams_stream_form_unordered_items_of_16_ms"] = streams_ms(3)   # the split kernel on a scene without GLASS
        ctx.set_option(B.OPT_STREAM_BATCH, 0)
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_AUTO)
        out["c2_streams_stream_form_ms"] = streams_ms(3)
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
        # --- the glass scene (GLASS extension), 1080p / 64 spp: tree walk (default) and stream form
        ctx.set_scene(*pkg.world.glass_scene())
        ctx.reset_stats()
        out["glass_1080p_64spp_tree_ms"] = streams_ms()
        st = ctx.stats()
        out["glass_tree_live_rays_per_sample"] = round(st["live_bounces"] / st["samples"], 3)
        out["glass_tree_dropped"] = st["stream_rays_dropped"]
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        out["glass_1080p_64spp_stream_form_ms"] = streams_ms(3)
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
        ctx.set_scene(sp, pl)
        # --- C3: 3840x2160, 256 spp, 8 bounces
        w, h = 3840, 2160
        ctx.resize(w, h)
        ctx.init_output(0x5EED1234)
        ctx.render(cam, 8, 16); ctx.synchronize()
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.render(cam, 8, 256)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        out["c3_4k_256spp_ms"] = round(dt * 1e3, 2)
        out["c3_Msamples_s"] = round(w * h * 256 * 8 / dt / 1e6, 1)
        out["c3_algorithmic_GBs"] = round(w * h * 256 * 56 / dt / 1e9, 1)
        out["c3_live_fraction"] = round(st["live_bounces"] / st["nominal_bounces"], 4)
    # --- C5 per part: the glass scene at 3840x2160 / 512 spp on one of 8 row-stripe parts (tree walk)
    with pkg.Context(0) as ctx:
        ctx.set_scene(*pkg.world.glass_scene())
        ctx.set_partition(10, 8, 3)
        ctx.resize(3840, 2160)
        ctx.init_output(0x5EED1234)
        ctx.render(cam, 8, 32, pkg.STREAMS); ctx.synchronize()
        ctx.init_output(0x5EED1234)
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.render(cam, 8, 512, pkg.STREAMS)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        out["c5_part_of_8_4k_512spp_ms"] = round(dt * 1e3, 2)
        out["c5_part_live_Grays_s"] = round(st["live_bounces"] / dt / 1e9, 2)
        out["c5_part_dropped"] = st["stream_rays_dropped"]
    # --- C0, the reference's own configuration: 800x600, mainScene (7 primitives), limit 15 -- per resident call of 1 spp (what
    #     `compileFor`'s closure does per iteration) and of 30 spp (computationLoop's smallest batch, app/Main.hs:209-211)
    with pkg.Context(0) as ctx:
        ctx.set_scene(*pkg.world.main_scene())
        ctx.resize(800, 600)
        ctx.init_output(0x5EED1234)
        for n_spp, key in ((1, "c0_800x600_limit15_1spp_call_us"), (30, "c0_800x600_limit15_30spp_call_us")):
            for _ in range(20):
                ctx.render(cam, 15, n_spp)
            ctx.synchronize()
            t0 = time.perf_counter()
            reps = 200
            for _ in range(reps):
                ctx.render(cam, 15, n_spp)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[key] = round(dt * 1e6, 1)
        out["c0_Msamples_s_at_30spp"] = round(800 * 600 * 30 * 15 / (out["c0_800x600_limit15_30spp_call_us"] * 1e-6) / 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
