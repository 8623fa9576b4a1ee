#!/usr/bin/env python3
"""c0_calls.py [lib.so] -- the reference's own configuration (800x600, mainScene, limit 15): microseconds per call of
  * the resident ptmi_render with 1, 2, 4, 8 and 30 samples (computationLoop batches at least 30, app/Main.hs:209-211);
  * the COMPATIBLE closure, ptmi_render1: seven host planes in, seven out, one sample (what `compileFor` was built on until 0.5);
  * the CHAINED closure, ptmi_render1_chained, one sample per call, nothing fetched:
      - `released_at_once`: the input token released right after the call (a C++ handle's destructor);
      - `released_32_late`: released 32 calls later (a Haskell finalizer: Scene.HIP runs a minor collection every 32 calls);
      - `consumed`: PTMI_CHAIN_CONSUME, rendered in place;
      - `never_released`: nobody releases -- beyond PTMI_OPT_CHAIN_SLOTS states every call moves the oldest to the host (the safety net);
  * the chained closure with the three colour planes fetched every 30 calls (graphicsLoop reading at the compute loop's batch rate).
One JSON object."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
if len(sys.argv) > 1:
    pkg.binding._lib = None
    pkg.binding.load_library(sys.argv[1])
cam = pkg.world.initial_camera()
W, H, LIMIT = 800, 600, 15
out = {"build_id": pkg.load_library().build_id, "config": "C0: 800x600, mainScene, limit 15, render Inline"}


def per_call(fn, calls, sync):
    t0 = time.perf_counter()
    for k in range(calls):
        fn(k)
    sync()
    return round((time.perf_counter() - t0) / calls * 1e6, 1)


with pkg.Context(0) as ctx:
    ctx.set_scene(*pkg.world.main_scene())
    ctx.resize(W, H)
    ctx.init_output(1)
    res = {}
    for n in (1, 2, 4, 8, 30):
        for _ in range(50):
            ctx.render(cam, LIMIT, n)
        ctx.synchronize()
        res["%d_spp_us" % n] = per_call(lambda k: ctx.render(cam, LIMIT, n), 200, ctx.synchronize)
    out["resident"] = res

    planes = list(ctx.download_state())
    for _ in range(5):
        planes = list(ctx.render1(cam, LIMIT, W, H, planes))
    state = {"p": planes}

    def compat(_k):
        state["p"] = list(ctx.render1(cam, LIMIT, W, H, state["p"]))
    out["compatible_render1_us"] = per_call(compat, 50, ctx.synchronize)

    def chained(lag, consume=False, calls=400, fetch_every=0):
        tok = ctx.chain_init_output(W, H, 1)
        toks = [tok]
        for _ in range(20):                                   # warm-up: blocks allocated, dispatch order recorded
            t, _f = ctx.render1_chained(cam, LIMIT, W, H, toks[-1], consume=consume)
            if not consume:
                ctx.chain_release(toks[-1])
            toks = [t]
        ctx.synchronize()
        base = ctx.chain_info()

        def call(k):
            want = ("r", "g", "b") if fetch_every and (k + 1) % fetch_every == 0 else ()
            t, _f = ctx.render1_chained(cam, LIMIT, W, H, toks[-1], consume=consume, fetch=want)
            toks.append(t)
            if consume:
                toks.pop(0)
            elif lag is not None and len(toks) > lag + 1:
                ctx.chain_release(toks.pop(0))
        us = per_call(call, calls, ctx.synchronize)
        info = ctx.chain_info()
        for t in toks:
            ctx.chain_release(t)
        return {"us_per_call": us, "evictions": info["evictions"] - base["evictions"], "states_held_at_the_end": info["states_on_device"] + info["states_on_host"],
                "device_slots": info["device_slots"]}
    out["chained"] = {"released_at_once": chained(0), "released_32_late": chained(32), "consumed": chained(None, consume=True),
                      "never_released": chained(None, calls=150), "released_at_once_colour_fetched_every_30_calls": chained(0, fetch_every=30)}
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.download_color()
    out["download_colour_planes_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
out["chained_vs_resident_1_spp"] = round(out["chained"]["released_32_late"]["us_per_call"] / out["resident"]["1_spp_us"], 2)
out["compatible_vs_chained"] = round(out["compatible_render1_us"] / out["chained"]["released_32_late"]["us_per_call"], 1)
print(json.dumps(out))
