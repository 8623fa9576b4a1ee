#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Msamples/s, paths x bounces).

A step = one pass of `render Inline` over the C2 workload: 1920x1080 pixels PER GPU of the
16-primitive scene, bounce limit 8, 64 samples per pixel.  With N GPUs the image is 1920 x (1080*N),
row-stripe partitioned over the N ranks, so every rank launches exactly the N = 1 kernel shape
(per-GPU work fixed: weak scaling; `--weak spp` scales the sample count instead), followed by the
RCCL gather of the colour planes to rank 0, which overlaps the next step's render.
Inputs (state planes, scene) are resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (render_inline_kernel)
against HBM as BASELINE.json asks, with algorithmic bytes = 56 B per pixel-sample
(SURVEY.md 8d); `cpu_baseline` times the CPU oracle (a port of the reference's -fcpu path,
the real Accelerate build is not runnable) on the host cores, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

WIDTH, HEIGHT, SPP_PER_GPU, BOUNCE_LIMIT = 1920, 1080, 64, 8
SEED0 = 0x5EED1234
with open(os.path.join(ROOT, "BASELINE.json")) as _f:
    METRIC = json.load(_f)["metric"]
BYTES_PER_PIXEL_SAMPLE = 56.0        # 2 x (3 x f32 + 4 x u32): read + write of one `render` call
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def cpu_baseline(pkg, spheres, planes, cam, budget_s=15.0):
    """Time the oracle (OpenMP over rows, all host cores) on a bounded sample of the same workload."""
    import numpy as np
    ora = graft.load_oracle()
    ora.build()
    threads = ora.max_threads()
    seeds = ora.gen_seeds(SEED0, 0, WIDTH * HEIGHT)
    start = [np.zeros((HEIGHT, WIDTH), np.float32) for _ in range(3)] + [s.reshape(HEIGHT, WIDTH) for s in seeds]
    t0 = time.perf_counter()
    ora.render_inline(spheres, planes, cam, WIDTH, HEIGHT, BOUNCE_LIMIT, 1, start, n_threads=threads)
    t1 = time.perf_counter() - t0
    spp = max(1, min(4 * SPP_PER_GPU, int(budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    _, live = ora.render_inline(spheres, planes, cam, WIDTH, HEIGHT, BOUNCE_LIMIT, spp, start, n_threads=threads)
    dt = time.perf_counter() - t0
    nominal = WIDTH * HEIGHT * spp * BOUNCE_LIMIT
    # per-core figure: a quarter of the rows, one sample, one thread (a few seconds)
    q = max(1, HEIGHT // 4)
    t0 = time.perf_counter()
    ora.render_inline(spheres, planes, cam, WIDTH, q, BOUNCE_LIMIT, 1, [a[:q] for a in start], n_threads=1)
    one = WIDTH * q * BOUNCE_LIMIT / (time.perf_counter() - t0) / 1e6
    return {"value": round(nominal / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "single_thread_value": round(one, 3),
            "sample": "%dx%d, scene S16, limit %d, %d spp (the GPU step is %d), %.1f s, C oracle (oracle/pt_oracle.c) with OpenMP"
                      % (WIDTH, HEIGHT, BOUNCE_LIMIT, spp, SPP_PER_GPU, dt),
            "live_fraction": round(live / nominal, 4)}


def load_traffic():
    """HBM bytes per launch from the committed PMC profile of this workload, if any (else null)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("hbm_bytes_per_launch")
    except Exception:
        return None


def load_valu_accounting(kernel_ms):
    """The VALU side of the story (the kernel's real bound), from the committed PMC summary of this workload:
    wave-instructions per launch and active lanes per instruction are properties of the code, not of the run; the
    issue rate they imply is computed with THIS run's kernel time.  None if the profile is missing."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)["derived"]
        instr = float(d["valu_wave_instr_per_launch"])
        simd_cycles = 1024 * kernel_ms * 1e-3 * 2.4e9            # 256 CUs x 4 SIMDs at the 2.4 GHz the SQ counters show
        return {"wave_instr_per_launch": round(instr), "active_lanes_per_instr": round(float(d["avg_active_lanes_per_valu_instr"]), 2),
                "simd_cycles_per_instr": round(simd_cycles / instr, 3),
                "note": "measured issue cost of the mix: 2 cycles (f32 add/mul, int) to 4 (fma, f64, cmp, cndmask, cvt) -- the VALU pipes are saturated",
                "source": "profiles/r01_pmc_summary.json"}
    except Exception:
        return None


def main():
    global WIDTH, HEIGHT, SPP_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (DESIGN.md); 0 = default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stripe-rows", type=int, default=0, help="0 = largest stripe <= 8 rows that deals every rank the same number of rows")
    ap.add_argument("--width", type=int, default=WIDTH, help="experiments only: the headline number is the default C2 workload")
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP_PER_GPU)
    ap.add_argument("--scene", choices=["s16", "main", "glass"], default="s16")
    ap.add_argument("--algorithm", choices=["inline", "streams"], default="inline")
    ap.add_argument("--weak", choices=["rows", "spp"], default="rows", help="what grows with the GPU count")
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP_PER_GPU = args.width, args.height, args.spp

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
    from haskell_path_tracer_amd.parallel import ColorGatherer, StripePartition

    spheres, planes = {"s16": pkg.world.scene16, "main": pkg.world.main_scene, "glass": pkg.world.glass_scene}[args.scene]()
    algorithm = pkg.INLINE if args.algorithm == "inline" else pkg.STREAMS
    cam = pkg.world.initial_camera()
    spp = SPP_PER_GPU * (world if args.weak == "spp" else 1)
    if args.weak == "rows":
        HEIGHT = HEIGHT * world                      # 1080 rows per GPU

    if args.stripe_rows <= 0:
        args.stripe_rows = next((s for s in range(8, 0, -1) if HEIGHT % (s * world) == 0), 8)
    ctx = pkg.Context(local_rank)
    ctx.set_scene(spheres, planes)
    part = StripePartition(HEIGHT, world, rank, args.stripe_rows)
    if world > 1:
        ctx.set_partition(args.stripe_rows, world, rank)
    ctx.resize(WIDTH, HEIGHT)
    assert ctx.local_rows == part.local_rows
    color = torch.zeros((3, ctx.local_rows, WIDTH), dtype=torch.float32, device="cuda")
    state = torch.zeros((4, ctx.local_rows, WIDTH), dtype=torch.int32, device="cuda")
    ctx.bind_torch(color, state)
    # One side stream carries the kernels, the torch events that time them and the RCCL gather
    # (the legacy NULL stream would make ptmi fall back to its own stream: include/ptmi.h).
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    ctx.set_variant(args.variant)
    ctx.init_output(SEED0)

    gather = ColorGatherer(part, WIDTH, color.dtype, color.device, dst=0) if world > 1 else None

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
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        ctx.render(cam, BOUNCE_LIMIT, spp, algorithm)
        torch.cuda.synchronize()
    if world > 1:
        gather.overlapped(color)                     # first collective = communicator set-up; not a warm-up step
        gather.wait()
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

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
        tot = torch.tensor([stats["live_bounces"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        live_total = int(tot[0])
    else:
        live_total = stats["live_bounces"]

    is_c2 = (WIDTH, HEIGHT // (world if args.weak == "rows" else 1), SPP_PER_GPU, args.scene, args.algorithm) == (1920, 1080, 64, "s16", "inline")
    if rank == 0:
        nominal_per_step = WIDTH * HEIGHT * spp * BOUNCE_LIMIT          # whole job, all ranks
        value = nominal_per_step * args.steps / elapsed / 1e6
        # dominant kernel, per launch on one GPU: pixels held x spp x 56 B / launch duration
        alg_bytes = ctx.local_rows * WIDTH * spp * BYTES_PER_PIXEL_SAMPLE
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d image, %d spp per step, bounce limit 8, scene %s "
                                   "(%d spheres + %d planes), render %s, seeds from seed0=0x5EED1234"
                                   % (("C2" if world == 1 else "C2 per GPU (1920x1080 pixels each, weak scaling by %s)" % args.weak) if is_c2 else "experiment",
                                      WIDTH, HEIGHT, spp, args.scene.upper(),
                                      len(spheres), len(planes), args.algorithm.capitalize()),
                       "width": WIDTH, "height": HEIGHT, "spp_per_step": spp, "bounce_limit": BOUNCE_LIMIT,
                       "primitives": int(len(spheres) + len(planes)),
                       "parallelism": "row stripes of %d rows over %d GPU(s)%s"
                                      % (args.stripe_rows, world, " + RCCL gather of colour planes" if world > 1 else ""),
                       "variant": args.variant},
            "live_bounce_fraction": round(live_total / (nominal_per_step * args.steps), 4),
            "live_Mbounces_per_s": round(live_total / elapsed / 1e6, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": load_traffic() if (world == 1 and is_c2) else None,
                         "kernel": "render_inline_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "HBM is the bound BASELINE.json names; the kernel is f32/f64 VALU-bound (DESIGN.md)",
                         "valu": load_valu_accounting(kernel_ms) if (world == 1 and is_c2) else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, spheres, planes, cam)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
