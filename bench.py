#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Msamples/s, paths x bounces).

A step = one pass of `render Inline` over one batch of synthetic input, state resident in HBM.

  N = 1 (default)        C2 = BASELINE.json configs[1]: 1920x1080, 64 spp, bounce limit 8, scene S16.
  N > 1 (default)        STRONG scaling on C4 = configs[3]: ONE fixed 3840x2160 image at 1024 spp, row stripes
                         dealt round-robin to the N ranks, RCCL gather of the colour planes to rank 0 every step
                         (overlapping the next step's render).  `--scaling strong --gpus 1` runs C4 on one GPU.
  --scaling weak         the round-1 mode: every rank renders 1920x1080 pixels x 64 spp of a 1920 x (1080 N)
                         image (`--weak spp` grows the sample count instead).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel against HBM as BASELINE.json asks, with
ALGORITHMIC bytes = 56 B per pixel-sample (what one `render` call per sample would move); because the
sample loop lives inside the kernel the PHYSICAL traffic is 56 B per pixel per launch, reported beside it
(`physical_*`), and the kernel's real bound -- VALU issue -- is priced in `roofline.valu`.  `cpu_baseline` times the
CPU oracle (a port of the reference's -fcpu path; the real Accelerate build is not runnable) on the host cores the
process may use, on a bounded sample of the same image.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

SEED0 = 0x5EED1234
BOUNCE_LIMIT = 8
C2 = dict(name="C2", width=1920, height=1080, spp=64)          # BASELINE.json configs[1]
C4 = dict(name="C4", width=3840, height=2160, spp=1024)        # BASELINE.json configs[3]
with open(os.path.join(ROOT, "BASELINE.json")) as _f:
    METRIC = json.load(_f)["metric"]
BYTES_PER_PIXEL_SAMPLE = 56.0        # 2 x (3 x f32 + 4 x u32): read + write of one `render` call
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def usable_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    cores = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return cores, affinity, quota


def cpu_baseline(pkg, spheres, planes, cam, width, height, budget_s=12.0):
    """Time the oracle (OpenMP over rows) on a bounded sample of the SAME image the GPU step renders:
    every row for the threaded figure, every 4th row of the same image for the single-thread figure."""
    import numpy as np
    ora = graft.load_oracle()
    ora.build()
    cores, affinity, quota = usable_cores()
    omp_max = ora.max_threads()
    seeds = ora.gen_seeds(SEED0, 0, width * height)
    start = [np.zeros((height, width), np.float32) for _ in range(3)] + [s.reshape(height, width) for s in seeds]
    t0 = time.perf_counter()
    ora.render_inline(spheres, planes, cam, width, height, BOUNCE_LIMIT, 1, start, n_threads=cores)
    t1 = time.perf_counter() - t0
    spp = max(1, min(256, int(budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    _, live = ora.render_inline(spheres, planes, cam, width, height, BOUNCE_LIMIT, spp, start, n_threads=cores)
    dt = time.perf_counter() - t0
    nominal = width * height * spp * BOUNCE_LIMIT
    # per-core figure: rows 0, 4, 8, ... of the same image (same camera, same rays), one thread
    rows = np.arange(0, height, 4, dtype=np.int32)
    band = [a[rows] for a in start]
    spp1 = 2
    t0 = time.perf_counter()
    ora.render_inline(spheres, planes, cam, width, height, BOUNCE_LIMIT, spp1, band, n_threads=1, rows=rows)
    one = width * len(rows) * spp1 * BOUNCE_LIMIT / (time.perf_counter() - t0) / 1e6
    value = nominal / dt / 1e6
    return {"value": round(value, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "single_thread_value": round(one, 3), "parallel_speedup": round(value / one, 2),
            "affinity_cpus": affinity, "cgroup_cpu_quota": quota, "omp_get_max_threads": omp_max,
            "sample": "%dx%d, scene S16, limit %d, %d spp (the GPU step is the same image), %.1f s, C oracle "
                      "(oracle/pt_oracle.c) with OpenMP dynamic row chunks on %d threads; single-thread figure on "
                      "every 4th row of the same image, %d spp" % (width, height, BOUNCE_LIMIT, spp, dt, cores, spp1),
            "live_fraction": round(live / nominal, 4)}


def load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def load_traffic():
    """HBM bytes per launch of the C2 kernel from the committed PMC profile (profiles/traffic.json), else None."""
    t = load_json("traffic.json")
    return t.get("hbm_bytes_per_launch") if t else None


def valu_accounting(pkg, kernel_ms):
    """The kernel's real bound.  The instruction mix and the cycle count are properties of the code and the workload, not
    of the run (two boxes of the pool at 2.28 and 2.36 GHz spent the same cycles): profiles/rNN_valu_roofline.json holds,
    for the C2 launch, the VALU wave-instructions per launch by category (PMC), the issue cycles their class costs assign
    to that mix (a lower bound), the SIMD cycles the launch took (GRBM_GUI_ACTIVE / 8 XCDs) and the active-lane fraction.
    frac = issue cycles needed / SIMD cycles taken, both from that profile; `stale` says whether the compiled CODE has
    changed since the profile was taken (the profile names the ptmi_build_id() of the binary it was taken on; comment edits keep
    that id: _build.code_id); implied_clock_ghz = the shader clock this run's launch time implies
    for the same cycle count -- outside 2.1-2.5 GHz the profile no longer describes the binary."""
    d, name = None, None
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        d = load_json("%s_valu_roofline.json" % tag)
        if d:
            name = "profiles/%s_valu_roofline.json" % tag
            break
    if not d:
        return None
    try:
        measured = float(d["measured_cycles_in_profile"])
        implied = measured / (d["n_simds"] * kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        now = pkg.load_library().build_id.split("+")[0]
        return {"min_issue_cycles": round(d["min_issue_cycles"]), "measured_cycles": round(measured),
                "frac": round(d["min_issue_cycles"] / measured, 4),
                "frac_with_v_fma_f32_at_2_cycles": d.get("frac_with_v_fma_f32_at_2_cycles"),
                "frac_with_v_fma_f32_as_measured": d.get("frac_with_v_fma_f32_as_measured"),
                "frac_priced_with_measured_opcode_costs": round(d["priced_with_measured_rates"]["frac"], 4),
                "active_lane_frac": round(d["active_lane_frac"], 4),
                "valu_wave_instr_per_launch": round(d["valu_wave_instr_per_launch"]),
                "avg_issue_cycles_per_instr": round(d["avg_issue_cycles_per_instr"], 3),
                "implied_clock_ghz": round(implied, 3) if implied else None,
                "profile_clock_ghz": d["clock_ghz"], "source": name,
                "profile_build_id": d.get("build_id", d.get("source_hash")), "build_id_now": now,
                "stale": d.get("build_id", d.get("source_hash")) != now}
    except Exception:
        return None


def streams_rooflines(pkg):
    """{workload key: the VALU issue accounting of its kernel} from the newest profiles/rNN_valu_roofline_streams.json
    (tools/valu_roofline.py streams TAG: the accounting of `roofline.valu`, for the Streams kernels), with `stale` by the same hash rule."""
    for tag in ("r06", "r05", "r04", "r03"):
        d = load_json("%s_valu_roofline_streams.json" % tag)
        if d:
            now = pkg.load_library().build_id.split("+")[0]
            return {key: {"kernel": a["kernel"], "valu_issue_frac": a["frac_in_profile"], "valu_issue_frac_priced_with_measured_opcode_costs": a["priced_with_measured_rates"]["frac"],
                          "active_lane_frac": a["active_lane_frac"], "simd_cycles_per_valu_instr": a["measured_simd_cycles_per_instr"],
                          "valu_wave_instr_per_call": round(a["valu_wave_instr_per_launch"]), "hbm_MB_per_call": a.get("hbm_MB_per_call"),
                          "kernel_us_in_profile": a["kernel_us_in_profile"], "source": a["source"], "profile_build_id": a.get("build_id", a.get("source_hash")),
                          "stale": a.get("build_id", a.get("source_hash")) != now} for key, a in d.items() if "refused" not in a}
    return {}


def also_measurements(pkg, torch):
    """The other BASELINE.json configs and the Streams forms, on this GPU, after the headline: 3 to 5 timed steps each (the mean is
    reported) behind a short warm-up (clock ramp, recorded dispatch order: it is rebuilt before launch 1, 2, 4, 8, ... of a context, so nine
    warm-up launches -- every Streams entry has them -- leave the timed ones without a rebuild), resident state, one context each on a
    stream of its own.
    kernel_ms: HIP events around the launches on the launch stream (ptmi_set_timing)."""
    B = pkg.binding
    cam = pkg.world.initial_camera()
    scenes = {"s16": pkg.world.scene16(), "main": pkg.world.main_scene(), "glass": pkg.world.glass_scene()}
    out = []
    rooflines = streams_rooflines(pkg)

    def run(name, scene, width, height, spp, limit, algorithm, part_of=0, stripe=10, stream_form=False, warm=3, steps=3, note=None, profile=None, part=0):
        sp, pl = scenes[scene]
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            if part_of > 1:
                c.set_partition(stripe, part_of, part)
            c.resize(width, height)
            c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if stream_form else B.FORM_PIXEL)     # (each line names its form: AUTO would pick the stream form for a glass part)
            c.init_output(SEED0)
            c.set_timing(True)
            for _ in range(warm):
                c.render(cam, limit, spp, algorithm)
            c.synchronize()
            wall, dev = [], []
            for _ in range(steps):
                t0 = time.perf_counter()
                c.render(cam, limit, spp, algorithm)
                c.synchronize()
                wall.append((time.perf_counter() - t0) * 1e3)
                dev.append(c.stats()["last_render_ms"])
            rows = c.local_rows
        ms, kms = sum(wall) / len(wall), sum(dev) / len(dev)
        rec = {"workload": name, "ms_per_step": round(ms, 4), "kernel_ms": round(kms, 4), "steps": steps,
               "algorithmic_GBps": round(rows * width * spp * BYTES_PER_PIXEL_SAMPLE / (kms * 1e-3) / 1e9, 1) if kms > 0 else None}
        if algorithm == pkg.INLINE:
            rec["Msamples_per_s"] = round(rows * width * spp * limit / (ms * 1e-3) / 1e6, 1)
        if note:
            rec["note"] = note
        if profile and profile in rooflines:                 # the kernel's real bound, from the committed PMC profile of this workload
            rec["roofline"] = dict(rooflines[profile], bound="valu issue")
        out.append(rec)

    run("C0: 800x600, mainScene, limit 15, render Inline, 1 spp per call (the reference's own configuration; compileFor's closure)", "main", 800, 600, 1, 15, pkg.INLINE, warm=40, steps=20)
    run("C0 at 30 spp per call (the reference's batch size, app/Main.hs:209-211)", "main", 800, 600, 30, 15, pkg.INLINE, warm=20, steps=10)

    def closure(name, chained, calls, note):
        """C0 through compileFor's closure itself -- one call = one sample, RenderResult in, RenderResult out: `chained` on
        ptmi_render1_chained (the result stays on the device under a token, the next call finds it there; the input token is released 32
        calls late, as a Haskell finalizer would), else on ptmi_render1 (seven host planes each way per call).  Wall clock per call over
        `calls` calls, the queue drained at the end."""
        sp, pl = scenes["main"]
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            if chained:
                toks = [c.chain_init_output(800, 600, SEED0)]
                def one():
                    toks.append(c.render1_chained(cam, 15, 800, 600, toks[-1])[0])
                    if len(toks) > 33:
                        c.chain_release(toks.pop(0))
            else:
                c.resize(800, 600)
                c.init_output(SEED0)
                state = {"p": list(c.download_state())}
                def one():
                    state["p"] = list(c.render1(cam, 15, 800, 600, state["p"]))
            for _ in range(40 if chained else 5):
                one()
            c.synchronize()
            # three timed blocks, the median reported, Python's cyclic collector off: in this process -- torch's heap beside it -- a full
            # collection takes ~40 ms, the loop allocates a few containers per call, and one collection inside a block of calls that
            # cost 30 us each multiplied the block's figure by ten (tools/chain_variance.py: always at the same call, gone with gc off)
            import gc
            blocks = []
            gc.collect(); gc.disable()
            try:
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(calls):
                        one()
                    c.synchronize()
                    blocks.append((time.perf_counter() - t0) * 1e3 / calls)
            finally:
                gc.enable()
            ms = sorted(blocks)[1]
            info = c.chain_info() if chained else None
        rec = {"workload": name, "ms_per_step": round(ms, 4), "kernel_ms": None, "steps": calls, "blocks_ms_per_call": [round(b, 4) for b in blocks], "note": note,
               "Msamples_per_s": round(800 * 600 * 15 / (ms * 1e-3) / 1e6, 1)}
        if info:
            rec["chain"] = {k: info[k] for k in ("renders_chained", "renders_uploaded", "evictions", "states_on_device", "states_on_host")}
        out.append(rec)
    closure("C0 through the compatible closure, chained (ptmi_render1_chained: compileFor's pure type, results left on the device; what haskell/patches/Main.hs.diff wires)",
            True, 150, "wall clock per closure call (median of three blocks); nothing is fetched (graphicsLoop's read of three colour planes costs ~0.27 ms when it happens)")
    closure("C0 through the compatible closure, copying (ptmi_render1: seven host planes in and out per call; the closure until 0.5)",
            False, 12, "wall clock per closure call (median of three blocks), PCIe both ways inside")
    run("C3: 3840x2160, 256 spp, limit 8, S16, render Inline", "s16", 3840, 2160, 256, BOUNCE_LIMIT, pkg.INLINE)
    run("C4: 3840x2160, 1024 spp, limit 8, S16, render Inline, the whole image on one GPU", "s16", 3840, 2160, 1024, BOUNCE_LIMIT, pkg.INLINE, warm=2)
    run("C4, one part of 8 (10-row stripes): what one rank of the 8-GPU job renders", "s16", 3840, 2160, 1024, BOUNCE_LIMIT, pkg.INLINE, part_of=8)
    run("C5, one part of 8: glass scene, 3840x2160, 512 spp, render Streams, per-pixel tree walk (PTMI_FORM_PIXEL: deterministic, bit-exact against the oracle)", "glass", 3840, 2160, 512, BOUNCE_LIMIT, pkg.STREAMS, part_of=8, warm=9, profile="c5_tree")
    run("C5, one part of 8, stream ('wavefront') form: start-hit regions, graded passes, child rings (BASELINE configs[4]'s path; what PTMI_FORM_AUTO picks for a glass part at >= 256 spp)", "glass", 3840, 2160, 512, BOUNCE_LIMIT, pkg.STREAMS, part_of=8, stream_form=True, warm=9, profile="c5_stream")
    # the part that bounds the 8-GPU job: under the tree walk part 6 of the 10-row stripes is the slowest in every run of tools/part_bound.py (profiles/r05_c5_part.json)
    run("C5, the slowest part of 8 (part 6), tree walk", "glass", 3840, 2160, 512, BOUNCE_LIMIT, pkg.STREAMS, part_of=8, part=6, warm=9)
    run("C5, part 6 of 8, stream form", "glass", 3840, 2160, 512, BOUNCE_LIMIT, pkg.STREAMS, part_of=8, part=6, stream_form=True, warm=9)
    run("C2 through render Streams, per-pixel chain", "s16", 1920, 1080, 64, BOUNCE_LIMIT, pkg.STREAMS, warm=9, steps=5, profile="streams")
    run("C2 through render Streams, stream form", "s16", 1920, 1080, 64, BOUNCE_LIMIT, pkg.STREAMS, stream_form=True, warm=9, steps=5, profile="s16_stream")
    run("glass scene, 1920x1080, 64 spp, render Streams, tree walk", "glass", 1920, 1080, 64, BOUNCE_LIMIT, pkg.STREAMS, warm=9, steps=5, profile="glass_tree")
    run("glass scene, 1920x1080, 64 spp, render Streams, stream form", "glass", 1920, 1080, 64, BOUNCE_LIMIT, pkg.STREAMS, stream_form=True, warm=9, steps=5, profile="glass_stream")
    return out


def pick_stripe(height, world):
    """Stripe height that deals every rank the same number of rows, closest to the 8-row tile of the kernel."""
    for s in (8, 10, 6, 12, 5, 4, 15, 16, 3, 2, 1):
        if height % (s * world) == 0:
            return s
    return 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="auto: C2 on one GPU, strong scaling on C4 when --gpus > 1")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (DESIGN.md); 0 = default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-image", action="store_true",
                    help="N > 1: after the timed steps rank 0 renders the same samples in ONE context and compares the gathered image bit for bit")
    ap.add_argument("--no-precheck", action="store_true",
                    help="N > 1: skip the check BEFORE the timed steps (stripes rendered at a reduced sample count, gathered without overlap and "
                         "compared on rank 0 with one context's image, bit for bit) and the timing of that one gather")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` block (the other BASELINE configs after the headline, ~4 s)")
    ap.add_argument("--no-n1-reference", action="store_true", help="strong scaling: skip the one-GPU timing of the same image on rank 0")
    ap.add_argument("--stripe-rows", type=int, default=0, help="0 = a stripe that deals every rank the same number of rows")
    ap.add_argument("--width", type=int, default=0, help="experiments only: the headline numbers are the default workloads")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--scene", choices=["s16", "main", "glass"], default="s16")
    ap.add_argument("--algorithm", choices=["inline", "streams"], default="inline")
    ap.add_argument("--weak", choices=["rows", "spp"], default="rows", help="weak scaling: what grows with the GPU count")
    ap.add_argument("--part-of", type=int, default=0,
                    help="one GPU only: render part 0 of this many row-stripe parts of the image (what one rank of an N-GPU job does), "
                         "e.g. C5 per part: --scene glass --algorithm streams --width 3840 --height 2160 --spp 512 --part-of 8")
    ap.add_argument("--streams-form", choices=["auto", "stream", "pixel"], default="auto",
                    help="render Streams: the library's choice (auto: per-pixel kernels, but the stream form for a GLASS scene on one part of a "
                         "partitioned image at >= 256 spp), the stream ('wavefront') form, or the per-pixel kernels")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="experiments only: ptmi_set_option before the first render, by the binding's name without OPT_ (e.g. STREAM_PASS_GROUPS=2); repeatable")
    ap.add_argument("--library", default=None,
                    help="experiments only: a differently-flagged build of libptmi made from THESE sources (a diagnostic build of tools/traffic_terms.py); "
                         "the binding still refuses a library built from other sources")
    ap.add_argument("--ramp-spp", type=int, default=64,
                    help="samples per launch of the untimed clock ramp before the warm-up (capped at the workload's spp); 0 = the workload's "
                         "own spp, so that EVERY launch of the process is a launch of the workload -- what tools/pmc_kernels.sh passes: a "
                         "per-kernel mean of a profile must not mix two launch sizes")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libptmi has no CPU path")

    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    base = C4 if scaling == "strong" else C2
    width, height, spp_base = args.width or base["width"], args.height or base["height"], args.spp or base["spp"]
    named = (width, height, spp_base, args.scene, args.algorithm) == (base["width"], base["height"], base["spp"], "s16", "inline")
    if args.steps is None:
        args.steps = 50 if scaling == "weak" else 8          # a C4 step is ~230 ms on one GPU
    if args.warmup is None:
        args.warmup = 10 if scaling == "weak" else 2

    # Rehearsal on a one-GPU box (never the measured path): all ranks share cuda:0 and talk over gloo.
    rehearsal = os.environ.get("PTMI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    cdev = "cpu" if rehearsal else "cuda"          # where the small reduction tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = graft.load_package()
    pkg._build.build_lib()
    if args.library:
        pkg.binding.load_library(os.path.abspath(args.library))
    from haskell_path_tracer_amd.parallel import ColorGatherer, StripePartition

    spheres, planes = {"s16": pkg.world.scene16, "main": pkg.world.main_scene, "glass": pkg.world.glass_scene}[args.scene]()
    algorithm = pkg.INLINE if args.algorithm == "inline" else pkg.STREAMS
    cam = pkg.world.initial_camera()
    spp = spp_base
    if scaling == "weak":
        if args.weak == "spp":
            spp = spp_base * world
        else:
            height = height * world                  # `height` rows per GPU

    if args.stripe_rows <= 0:
        args.stripe_rows = pick_stripe(height, world)
    ctx = pkg.Context(local_rank)
    ctx.set_scene(spheres, planes)
    n_parts, my_part = (args.part_of, 0) if (world == 1 and args.part_of > 1) else (world, rank)
    if n_parts != world and args.stripe_rows == pick_stripe(height, world):
        args.stripe_rows = pick_stripe(height, n_parts)
    part = StripePartition(height, n_parts, my_part, args.stripe_rows)
    if n_parts > 1:
        ctx.set_partition(args.stripe_rows, n_parts, my_part)
    # which form of render Streams this run is in (PTMI_FORM_AUTO's rule, include/ptmi.h: the stream form for GLASS on a part at >= 256 spp)
    in_stream_form = args.algorithm == "streams" and (args.streams_form == "stream" or (
        args.streams_form == "auto" and args.scene == "glass" and n_parts > 1 and spp >= 256))
    ctx.resize(width, height)
    assert ctx.local_rows == part.local_rows
    color = torch.zeros((3, ctx.local_rows, width), dtype=torch.float32, device="cuda")
    state = torch.zeros((4, ctx.local_rows, width), dtype=torch.int32, device="cuda")
    ctx.bind_torch(color, state)
    # One side stream carries the kernels, the torch events that time them and the RCCL gather
    # (the legacy NULL stream would make ptmi fall back to its own stream: include/ptmi.h).
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    ctx.set_variant(args.variant)
    if args.streams_form != "auto":
        ctx.set_option(pkg.binding.OPT_STREAMS_FORM, pkg.binding.FORM_STREAM if args.streams_form == "stream" else pkg.binding.FORM_PIXEL)
    for item in args.option:
        name, value = item.split("=")
        ctx.set_option(getattr(pkg.binding, "OPT_" + name), int(value))
    ctx.init_output(SEED0)

    gather = ColorGatherer(part, width, color.dtype, color.device, dst=0) if world > 1 else None

    def step():
        ctx.render(cam, BOUNCE_LIMIT, spp, algorithm)
        if world > 1:
            return gather.overlapped(color)          # the gather of step k runs beside the render of step k + 1
        return None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, untimed and outside the W/K protocol: load the code object and let the device leave its idle
    # clock state (the first ~10 launches after start-up run ~7 % slower), so that W and K measure steady state
    # whatever their values.  State is re-initialised afterwards.
    ramp_spp = spp if args.ramp_spp <= 0 else min(spp, args.ramp_spp)
    t_ramp = time.perf_counter()
    ramp_launches = 0
    while time.perf_counter() - t_ramp < 0.3:
        ctx.render(cam, BOUNCE_LIMIT, ramp_spp, algorithm)
        torch.cuda.synchronize()
        ramp_launches += 1
    if world > 1:
        gather.overlapped(color)                     # first collective = communicator set-up; not a warm-up step
        gather.wait()
    # N > 1, before anything is timed: does the job produce the right image, and did RCCL see N ranks on N devices?  The stripes at a
    # reduced sample count, ONE gather that overlaps nothing (timed on its own: what xGMI delivers per peer), and on rank 0 the same
    # samples rendered by one context, compared bit for bit.
    collective = None
    if world > 1:
        ctx.init_output(SEED0)
        spp_check = min(spp, 16)
        ctx.render(cam, BOUNCE_LIMIT, spp_check, algorithm)
        fence()
        t_g = time.perf_counter()
        checked = gather(color)                      # on the render stream, nothing beside it
        fence()
        t_gather = time.perf_counter() - t_g
        tg = torch.tensor([t_gather], dtype=torch.float64, device=cdev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        mine = {"rank": rank, "cuda_device": int(torch.cuda.current_device()), "name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")), "rows": int(ctx.local_rows)}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        bytes_per_rank = 3 * part.max_rows() * width * 4
        collective = {"backend": dist.get_backend(), "world_size": int(dist.get_world_size()),
                      "distinct_devices": len({(d["cuda_device"], d["pci_bus_id"], d["uuid"]) for d in seen}),
                      "devices": seen, "collective": "torch.distributed.gather of the three colour planes to rank 0 (RCCL: every peer sends over its own xGMI link)",
                      "bytes_per_rank": bytes_per_rank, "gather_ms_not_overlapped": round(float(tg[0]) * 1e3, 4),
                      "GBps_per_peer": round(bytes_per_rank / float(tg[0]) / 1e9, 2) if float(tg[0]) > 0 else None,
                      "GBps_into_root": round(bytes_per_rank * (world - 1) / float(tg[0]) / 1e9, 2) if float(tg[0]) > 0 else None,
                      "note": "the timed steps overlap this gather with the next step's render; this figure is one gather alone (stitch into [3][H][W] on rank 0 included)"}
        if not args.no_precheck:
            ok = None
            if rank == 0:
                import numpy as np
                with pkg.Context(local_rank) as whole:
                    whole.set_scene(spheres, planes)
                    whole.resize(width, height)
                    whole.init_output(SEED0)
                    whole.render(cam, BOUNCE_LIMIT, spp_check, algorithm)
                    want = whole.download_color()
                got = checked.cpu().numpy()
                ok = all(np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in range(3))
            flag = [ok]
            dist.broadcast_object_list(flag, src=0)
            collective["gathered_image_equals_one_context_at_%d_spp" % spp_check] = bool(flag[0])
            if not flag[0]:
                raise SystemExit("bench.py: the gathered image of %d ranks differs from one context's image: nothing is timed" % world)
    ctx.init_output(SEED0)
    for _ in range(args.warmup):
        step()
    if world > 1:
        gather.wait()
    fence()
    ctx.reset_stats()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        ctx.render(cam, BOUNCE_LIMIT, spp, algorithm)
        ev[k][1].record()
        if world > 1:
            gather.overlapped(color)
    if world > 1:
        gather.wait()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    stats = ctx.stats()

    per_rank, render_only_ms, k_only = None, None, 0
    if world > 1:
        # what every rank did in the timed region, so that a scaling record explains itself: the slowest rank sets the step (MAX below),
        # imbalance = slowest rank's kernel time / the mean says how much of a missing speed-up is the stripes' doing
        mine = torch.tensor([elapsed, kernel_ms, float(ctx.local_rows), float(stats["live_bounces"])], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "elapsed_ms_per_step": round(float(v[0]) / max(args.steps, 1) * 1e3, 4), "kernel_ms": round(float(v[1]), 4),
                     "rows": int(v[2]), "live_bounces": int(v[3])} for r, v in enumerate(every)]
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
        tot = torch.tensor([stats["live_bounces"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        live_total = int(tot[0])
        # the same steps WITHOUT the gather (outside the timed region, same fences, MAX over ranks): what the overlap has to hide
        k_only = max(1, min(args.steps, 4))
        fence()
        t1 = time.perf_counter()
        for _ in range(k_only):
            ctx.render(cam, BOUNCE_LIMIT, spp, algorithm)
        fence()
        t_only = torch.tensor([(time.perf_counter() - t1) / k_only], dtype=torch.float64, device=cdev)
        dist.all_reduce(t_only, op=dist.ReduceOp.MAX)
        render_only_ms = float(t_only[0]) * 1e3
    else:
        live_total = stats["live_bounces"]

    # Strong scaling: the same whole image on ONE GPU (rank 0 alone, outside the timed region), so that the line
    # carries the one-GPU time of exactly this workload next to the N-GPU time.
    n1_ms = None
    if scaling == "strong" and world > 1 and rank == 0 and not args.no_n1_reference:
        with pkg.Context(local_rank) as whole:
            whole.set_scene(spheres, planes)
            whole.resize(width, height)
            whole.set_stream(stream.cuda_stream)
            whole.set_variant(args.variant)
            whole.init_output(SEED0)
            times = []
            for _ in range(3):                       # the second and third run in the recorded cost order
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                whole.render(cam, BOUNCE_LIMIT, spp, algorithm)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t1) * 1e3)
            n1_ms = min(times)
    image_equal = None
    if args.check_image and world > 1:
        # the image rank 0 holds after the last gather against the same samples rendered by ONE context: stripes, seeds from the
        # global pixel index, gather and reassembly change no bit
        final = gather.overlapped(color)
        gather.wait()
        torch.cuda.synchronize()
        if rank == 0:
            with pkg.Context(local_rank) as whole:
                whole.set_scene(spheres, planes)
                whole.resize(width, height)
                whole.init_output(SEED0)
                for _ in range(args.warmup + args.steps + k_only):      # every render since the last init_output, the render-only steps included
                    whole.render(cam, BOUNCE_LIMIT, spp, algorithm)
                want = whole.download_color()
            got = final.cpu().numpy()
            import numpy as np
            image_equal = all(np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in range(3))
    if world > 1:
        dist.barrier()

    if rank == 0:
        rows_total = ctx.local_rows if n_parts != world else height      # --part-of: the job is this part only
        nominal_per_step = width * rows_total * spp * BOUNCE_LIMIT       # whole job, all ranks
        value = nominal_per_step * args.steps / elapsed / 1e6
        # dominant kernel, per launch on one GPU: pixels held x spp x 56 B / launch duration
        alg_bytes = ctx.local_rows * width * spp * BYTES_PER_PIXEL_SAMPLE
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        is_c2 = named and scaling == "weak" and world == 1
        traffic = load_traffic() if is_c2 else None
        shape = (width, height, spp, args.scene, args.algorithm)
        if shape == (3840, 2160, 256, "s16", "inline"):
            config_name = "C3"                                            # BASELINE.json configs[2]
        elif shape == (3840, 2160, 512, "glass", "streams"):
            config_name = "C5 (%s)" % ("stream form" if in_stream_form else "per-pixel tree walk")   # configs[4]
        else:
            config_name = "experiment"
        if n_parts != world:
            config_name += ", one part of %d" % n_parts
        if scaling == "strong":
            workload = "%s: ONE %dx%d image at %d spp, row-striped over %d GPU(s) (strong scaling)" % (
                base["name"] if named else config_name, width, height, spp, world)
        else:
            workload = ("C2" if world == 1 else "C2 per GPU (1920x1080 pixels each, weak scaling by %s)" % args.weak) if named else config_name
            workload += ": %dx%d image, %d spp per step" % (width, height, spp)
        kernel_name = "render_inline_kernel" if args.algorithm == "inline" else \
            (("streams_split_kernel (stream form)" if args.scene == "glass" else "streams_pixels_kernel (stream form)") if in_stream_form else
             ("render_streams_tree_kernel" if args.scene == "glass" else "render_streams_kernel"))
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "achieved_is": "ALGORITHMIC-equivalent GB/s: 56 B per pixel-sample, what one `render` call per sample "
                                   "would move (SURVEY.md 8d); the fused sample loop moves 56 B per pixel per LAUNCH",
                    "physical_GBps": round(traffic / (kernel_ms * 1e-3) / 1e9, 2) if traffic and kernel_ms > 0 else None,
                    "physical_frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and kernel_ms > 0 else None,
                    "real_bound": "valu",
                    "valu": valu_accounting(pkg, kernel_ms) if is_c2 else None}
        out = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, bounce limit %d, scene %s (%d spheres + %d planes), render %s, seeds from seed0=0x5EED1234"
                                   % (workload, BOUNCE_LIMIT, args.scene.upper(), len(spheres), len(planes), args.algorithm.capitalize()),
                       "width": width, "height": height, "spp_per_step": spp, "bounce_limit": BOUNCE_LIMIT,
                       "primitives": int(len(spheres) + len(planes)),
                       "parallelism": "row stripes of %d rows over %d GPU(s)%s"
                                      % (args.stripe_rows, world, " + RCCL gather of colour planes" if world > 1 else ""),
                       "rows_per_gpu": ctx.local_rows, "variant": args.variant,
                       "part_of": args.part_of if n_parts != world else None,
                       "options": args.option or None},
            "live_bounce_fraction": round(live_total / (nominal_per_step * args.steps), 4) if args.algorithm == "inline" else None,
            "live_Mbounces_per_s": round(live_total / elapsed / 1e6, 1),
            "roofline": roofline,
            # which binary produced these numbers: the id linked into the loaded libptmi.so (ptmi_build_id: a hash over its compiled code) and
            # the id the sources beside it compile to now -- binding.open_library has already refused the library if they differ;
            # source_text_hash only says whether any byte of the sources (comments included) has moved since the link
            "binary_build_id": pkg.load_library().build_id, "code_id_now": pkg._build.code_id(),
            "source_text_hash": pkg._build.source_hash(), "binary_linked_from_text": pkg._build.read_source_hash(pkg._build.LIB),
            "ramp": {"spp_per_launch": ramp_spp, "launches": ramp_launches},
        }
        if collective is not None:
            collective["per_rank"] = per_rank
            kms = [r["kernel_ms"] for r in per_rank]
            collective["imbalance"] = round(max(kms) / (sum(kms) / len(kms)), 4) if sum(kms) > 0 else None
            collective["imbalance_is"] = "slowest rank's kernel_ms / mean kernel_ms over the ranks, timed region (1 = perfectly even stripes)"
            step_ms = elapsed / max(args.steps, 1) * 1e3
            g_ms = collective["gather_ms_not_overlapped"]
            collective["render_only_ms_per_step"] = round(render_only_ms, 4)
            collective["gather_hidden_frac"] = round(1.0 - (step_ms - render_only_ms) / g_ms, 4) if g_ms > 0 else None
            collective["gather_hidden_frac_is"] = ("1 - (ms_per_step with the overlapped gather - render_only_ms_per_step) / gather_ms_not_overlapped: "
                                                   "1 = the gather costs the step nothing, 0 = it costs as much as when nothing runs beside it "
                                                   "(figures of different loops: read it within their noise)")
            out["collective"] = collective
        if image_equal is not None:
            out["gathered_image_equals_one_context"] = bool(image_equal)
        if n1_ms is not None:
            out["one_gpu_same_workload"] = {"ms_per_step": round(n1_ms, 3),
                                            "value": round(nominal_per_step / (n1_ms * 1e-3) / 1e6, 1),
                                            "note": "the whole image on rank 0's GPU alone, best of 3, outside the timed region -- the one-GPU figure of THIS "
                                                    "workload (C4, strong scaling).  The N = 1 line of a scale series is the headline workload C2 (BASELINE "
                                                    "configs[1], 1080p / 64 spp), whose launches are 58 times shorter: a ratio of this line's value to that "
                                                    "line's compares two workloads; `python bench.py --gpus 1 --scaling strong` prints C4 on one GPU as a line of its own"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, spheres, planes, cam, width, height)
        if is_c2 and not args.no_also and not args.variant:
            t_also = time.perf_counter()
            out["also"] = also_measurements(pkg, torch)
            out["also_seconds"] = round(time.perf_counter() - t_also, 2)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
