#!/usr/bin/env python3
"""phase_stats.py -- where do the lanes go?  Builds a -DPTMI_PHASE_STATS copy of libptmi (diagnostic,
never the measured library), renders C2 once with render_inline_kernel and prints, per round of its
[shade A][shade B][trace C] loop, the fraction of lane-slots that did work, plus the wave-tail factor
(lane-iterations the waves paid for / lane-iterations needed)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def main():
    pkg = graft.load_package()
    out = "/tmp/libptmi_phase_stats.so"
    pkg._build.build_lib(out=out, extra_flags=["-DPTMI_PHASE_STATS", "-DPTMI_ABLATIONS"])
    pkg.binding._lib = None
    pkg.binding.load_library(out)
    sp, pl = pkg.world.scene16()
    w, h, spp = 1920, 1080, 64
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.resize(w, h)
        ctx.set_variant(int(os.environ.get("PTMI_PHASE_VARIANT", "13")))     # 13 = the default (8x8 tiles); 4 = row segments
        ctx.init_output(0x5EED1234)
        ctx.reset_stats()
        ctx.render(pkg.world.initial_camera(), 8, spp)
        c = ctx.debug_counters().astype(float)
        ctx.set_variant(1)
        ctx.init_output(0x5EED1234)
        ctx.reset_stats()
        ctx.render(pkg.world.initial_camera(), 8, spp)
        p = ctx.debug_counters().astype(float)
    print(json.dumps({"persistent_kernel": {
        "lane_slots_paid": p[1], "round_A_occupancy": p[2] / p[1], "round_B_occupancy": p[3] / p[1],
        "round_C_occupancy": p[4] / p[1], "useful_trace_slots": p[4]}}))
    raw = c.astype("uint32")
    cyc = [int(raw[8 + 2 * k]) | (int(raw[9 + 2 * k]) << 32) for k in range(3)]
    print(json.dumps({"wave_cycle_share": {"round_A": cyc[0] / sum(cyc), "round_B": cyc[1] / sum(cyc), "round_C_trace": cyc[2] / sum(cyc)}}))
    # second diagnostic build: per-sphere statistics (many atomics: never combined with the cycle stamps)
    out2 = "/tmp/libptmi_sphere_stats.so"
    pkg._build.build_lib(out=out2, extra_flags=["-DPTMI_SPHERE_STATS", "-DPTMI_ABLATIONS"])
    pkg.binding._lib = None
    pkg.binding.load_library(out2)
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.resize(w, h)
        ctx.set_variant(int(os.environ.get("PTMI_PHASE_VARIANT", "13")))
        ctx.init_output(0x5EED1234)
        ctx.reset_stats()
        ctx.render(pkg.world.initial_camera(), 8, spp)
        raw = ctx.debug_counters().astype("uint32")
    print(json.dumps({"sphere_tests": {"per_wave_tests": float(raw[16]), "fraction_taking_sqrt_path": float(raw[17]) / max(float(raw[16]), 1),
                                       "candidate_lanes_per_test": float(raw[18]) / max(float(raw[16]), 1),
                                       "active_lanes_per_test": float(raw[19]) / max(float(raw[16]), 1), "spp": spp},
                      "plane_tests": {"per_wave_tests": float(raw[20]), "fraction_taking_division_path": float(raw[21]) / max(float(raw[20]), 1)}}))
    # how often a WAVE runs each block of the loop in one C2 launch (64 spp): what tools/isa_other.py multiplies static instruction counts with
    print(json.dumps({"wave_block_executions": {"trips": c[24], "frozen_check": c[25], "frozen_finish": c[26], "restart": c[27], "shade": c[28], "trace": c[29],
                                                "sphere_tests": float(raw[16]), "sphere_sqrt_path": float(raw[17]), "plane_tests": float(raw[20]), "plane_division_path": float(raw[21]),
                                                "waves": w * h / 64.0}}))
    lane_iter, a, b, cc, paid = c[1], c[2], c[3], c[4], c[5]
    print(json.dumps({
        "lane_iterations_needed": lane_iter, "lane_iterations_paid_by_waves": paid,
        "wave_tail_factor": paid / lane_iter,
        "round_A_occupancy": a / lane_iter, "round_B_occupancy": b / lane_iter, "round_C_occupancy": cc / lane_iter,
        "iterations_per_sample": lane_iter / (w * h * spp),
        "frozen_shades_finished_cheaply_per_sample": c[6] / (w * h * spp),
        "note": "round A counts lanes entering the shade round (frozen finishes included); round B exists only in variant 18 (round 1's loop)"}))


if __name__ == "__main__":
    main()
