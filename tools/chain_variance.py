#!/usr/bin/env python3
"""chain_variance.py [torch] -- per-block wall time of the chained closure at C0 (800x600), blocks of 50 calls, 12 blocks per context:
[enqueue us per call, enqueue + drain us per call], for the copying form (input kept), the consuming form (no device copy) and the
resident call; `torch`: with torch imported and its CUDA context made first, as in bench.py; `nogc`: Python's cyclic collector off.
Round 6's finding: the ~40-ms stalls of the "keep" loop in a torch-laden process (one block of 50 calls at 700-900 us per call, always at
the same position) are full collections of Python's garbage collector in the MEASURING process -- the loop allocates a few containers per
call -- not the library: with `nogc` they are gone.  bench.py therefore times its closure lines with the collector off."""
import gc, sys, time, json
sys.path.insert(0, __file__.rsplit("/", 2)[0])
if "nogc" in sys.argv:
    gc.disable()
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init(); torch.zeros(1, device="cuda")
import __graft_entry__ as graft
pkg = graft.load_package()
cam = pkg.world.initial_camera()
sp, pl = pkg.world.main_scene()
out = {}
for mode in ("keep", "consume", "resident"):
    res = []
    for rep in range(3):
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.resize(800, 600); c.init_output(1)
            toks = [c.chain_init_output(800, 600, 1)]
            def one():
                if mode == "resident":
                    c.render(cam, 15, 1); return
                toks.append(c.render1_chained(cam, 15, 800, 600, toks[-1], consume=(mode == "consume"))[0])
                if mode == "consume": toks.pop(0)
                elif len(toks) > 33: c.chain_release(toks.pop(0))
            for _ in range(40): one()
            c.synchronize()
            per = []
            for blk in range(12):
                t0 = time.perf_counter()
                for _ in range(50): one()
                t1 = time.perf_counter()
                c.synchronize()
                t2 = time.perf_counter()
                per.append((round((t1 - t0) / 50 * 1e6, 1), round((t2 - t0) / 50 * 1e6, 1)))
            res.append(per)
    out[mode] = res
print(json.dumps(out))
