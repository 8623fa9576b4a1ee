"""Randomised parity campaign: random scenes (overlapping / tiny / huge / zero-radius spheres, un-normalised
plane normals, BRDF parameters outside their documented ranges, emissive everything), random cameras
(positions inside primitives, rotations beyond 120 rad, odd fields of view), random image shapes, bounce
limits and sample counts -- device vs oracle, BIT-EXACT on all seven planes, for both algorithms.  Seeded:
the same cases every run.  This is where a fast path that is only "almost always" equal to the literal
fold, square root or sin/cos would show."""
import os

import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("PTMI_FUZZ_CASES", "600"))        # a longer campaign: PTMI_FUZZ_CASES=30000 PTMI_FUZZ_SEED=...
SEED = int(os.environ.get("PTMI_FUZZ_SEED", "20260101"))
INLINE_VARIANTS = [0, 1, 0, 2, 0, 3, 0, 5, 0, 6, 0, 10, 0, 11, 0, 12, 0, 4, 0, 13, 0, 17, 0, 14]


def random_case(pkg, r):
    w = pkg.world
    ns, npl = int(r.integers(0, 9)), int(r.integers(0, 4))
    if ns + npl == 0:
        ns = 1
    spheres = np.zeros(ns, w.SPHERE_DTYPE)
    for i in range(ns):
        spheres["position"][i] = r.uniform(-12, 12, 3)
        spheres["radius"][i] = r.choice([0.0, 0.05, 0.5, 2.0, 7.0, 30.0]) * r.uniform(0.5, 1.5)
        spheres["color"][i] = r.uniform(0, 1.2, 3)
        spheres["illuminance"][i] = r.choice([0.0, 0.0, 1.0, 500.0])
        spheres["brdf_tag"][i] = r.integers(0, 2)
        spheres["brdf_param"][i] = r.choice([0.0, 0.3, 0.8, 1.0, 1.5, 3.0, -0.5])
    planes = np.zeros(npl, w.PLANE_DTYPE)
    for j in range(npl):
        planes["position"][j] = r.uniform(-15, 15, 3)
        n = r.normal(0, 1, 3)
        planes["direction"][j] = n * r.choice([1.0, 1.0, 0.3, 4.0]) / np.linalg.norm(n) if r.random() < 0.8 else (0.0, r.choice([-1.0, 1.0]), 0.0)
        planes["color"][j] = r.uniform(0, 1, 3)
        planes["illuminance"][j] = r.choice([0.0, 0.0, 2.0])
        planes["brdf_tag"][j] = r.integers(0, 2)
        planes["brdf_param"][j] = r.choice([0.0, 0.5, 0.9, 1.5])
    cam = w.camera(r.uniform(-10, 10, 3), r.uniform(-4, 4, 3) * r.choice([1.0, 1.0, 60.0]), int(r.choice([30, 60, 90, 120, 170])))
    width, height = int(r.integers(1, 90)), int(r.integers(1, 70))
    return spheres, planes, cam, width, height, int(r.choice([1, 2, 4, 8, 15])), int(r.integers(1, 4))


def test_random_scenes_inline_and_streams(ctx, actx, pkg, ora):
    r = np.random.default_rng(SEED)
    checked = 0
    for case in range(N_CASES):
        spheres, planes, cam, w, h, limit, spp = random_case(pkg, r)
        start = initial_planes(ora, w, h, seed0=int(r.integers(0, 2 ** 63)))
        variant = INLINE_VARIANTS[case % len(INLINE_VARIANTS)]     # half the cases on the default kernel, the rest spread over the others
        ic = ctx if variant in (0, 4, 5, 13, 17) else actx         # the other loop shapes live in the ablation library
        ic.set_scene(spheres, planes)
        ic.resize(w, h)
        ic.upload_state(*start)
        ic.reset_stats()
        ic.set_variant(variant)
        if case % 4 == 0 and spp >= 2:                             # two launches: the second runs in the cost order the first recorded
            ic.render(cam, limit, 1, pkg.INLINE)
            ic.render(cam, limit, spp - 1, pkg.INLINE)
        else:
            ic.render(cam, limit, spp, pkg.INLINE)
        ic.set_variant(0)
        got = ic.download_state()
        live_gpu = ic.stats()["live_bounces"]
        with np.errstate(all="ignore"):
            want, live = ora.render_inline(spheres, planes, cam, w, h, limit, spp, start)
        assert_planes_equal(got, want, "fuzz case %d inline, variant %d (%dx%d, %d+%d prims, limit %d, spp %d)" % (case, variant, w, h, len(spheres), len(planes), limit, spp))
        assert live_gpu == live
        if case % 3 == 0:
            if ic is not ctx:
                ctx.set_scene(spheres, planes)
                ctx.resize(w, h)
            ctx.upload_state(*start)
            stream_form = case % 2 == 1                            # the stream ("wavefront") form or the per-pixel form
            ctx.set_variant(9 if stream_form else 0)
            # (stream form: every other case cuts the pixels' sample chains into ordered passes of 1 to 3 samples -- hand-offs between
            # lanes through the planes; no result may depend on it)
            batch = int(r.choice([0, 1, 2, 3])) if stream_form else 0
            ctx.set_option(pkg.binding.OPT_STREAM_BATCH, batch)
            ctx.render(cam, limit, spp, pkg.STREAMS)
            ctx.set_option(pkg.binding.OPT_STREAM_BATCH, 0)
            ctx.set_variant(0)
            got = ctx.download_state()
            with np.errstate(all="ignore"):
                if stream_form:                                    # both forms share the 65 536-step safety cap
                    want = ora.render_streams_wavefront(spheres, planes, cam, w, h, 1 << 16, spp, start)[0]
                else:
                    want, _ = ora.render_streams(spheres, planes, cam, w, h, 1 << 16, spp, start)
            assert_planes_equal(got, want, "fuzz case %d streams (%s)" % (case, "stream form, items of %d samples" % batch if stream_form else "per-pixel form"))
        checked += 1
        if case % 5000 == 4999:
            print("fuzz: %d cases" % (case + 1), flush=True)
    assert checked == N_CASES


EXTREMES = np.array([0.0, -0.0, 1e-42, -1e-42, 1e-30, 1e30, -1e30, 3.4028235e38, np.inf, -np.inf, np.nan], np.float32)
N_EXTREME = int(os.environ.get("PTMI_FUZZ_EXTREME_CASES", "300"))


def poison(r, arr, rate):
    """Replace a fraction of the float fields of a record array by zeros, denormals, huge values, infinities, NaN."""
    for name in arr.dtype.names:
        if arr.dtype[name].base == np.float32:
            v = np.array(arr[name], np.float32).reshape(-1)
            hit = r.random(v.size) < rate
            v[hit] = r.choice(EXTREMES, int(hit.sum()))
            arr[name] = v.reshape(np.shape(arr[name]))
    return arr


def assert_planes_equal_up_to_nan_payload(got, want, what):
    """Bit-exact, except that any NaN equals any NaN (payload and sign of a NaN are not defined by the reference)."""
    for k, (a, b) in enumerate(zip(got, want)):
        ua, ub = np.asarray(a).reshape(-1).view(np.uint32), np.asarray(b).reshape(-1).view(np.uint32)
        same = ua == ub
        if k < 3:
            same |= np.isnan(np.asarray(a).reshape(-1)) & np.isnan(np.asarray(b).reshape(-1))
        if not same.all():
            bad = np.flatnonzero(~same)
            raise AssertionError("%s plane %d: %d of %d differ (first at %d: %#x vs %#x)" % (what, k, bad.size, ua.size, bad[0], ua[bad[0]], ub[bad[0]]))


def test_random_scenes_with_non_finite_and_denormal_numbers(ctx, pkg, ora):
    """The same campaign with zeros, signed zeros, denormals, 1e30-sized values, infinities and NaNs sprinkled over
    the scene, the camera and the accumulated colour: every comparison and selection has to go the reference's way
    (NaN keys in the fold, division by zero in the plane test, sqrt of negative and denormal numbers, sin/cos of
    huge and non-finite angles).  Bit-exact up to NaN payloads."""
    r = np.random.default_rng(SEED + 1)
    for case in range(N_EXTREME):
        spheres, planes, cam, w, h, limit, spp = random_case(pkg, r)
        rate = float(r.choice([0.02, 0.1, 0.3]))
        spheres, planes = poison(r, spheres, rate), poison(r, planes, rate)
        if r.random() < 0.3:
            cam = poison(r, np.array(cam, copy=True), 0.3)
        start = list(initial_planes(ora, w, h, seed0=int(r.integers(0, 2 ** 63))))
        if r.random() < 0.3:
            for k in range(3):
                v = start[k].reshape(-1).copy()
                hit = r.random(v.size) < 0.05
                v[hit] = r.choice(EXTREMES, int(hit.sum()))
                start[k] = v.reshape(start[k].shape)
        ctx.set_scene(spheres, planes)
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.reset_stats()
        ctx.render(cam, limit, spp, pkg.INLINE)
        got = ctx.download_state()
        live_gpu = ctx.stats()["live_bounces"]
        with np.errstate(all="ignore"):
            want, live = ora.render_inline(spheres, planes, cam, w, h, limit, spp, start)
        what = "extreme case %d (%dx%d, %d+%d prims, limit %d, spp %d)" % (case, w, h, len(spheres), len(planes), limit, spp)
        assert_planes_equal_up_to_nan_payload(got, want, what + " inline")
        assert live_gpu == live, what
        # Streams has no bounce limit (Trace.hs:166-170): a path whose throughput is NaN or infinite runs into the
        # 65 536-step cap, so only small images go through it here (the oracle walks those steps on one thread)
        if case % 3 == 0 and w * h * spp <= 150:
            with np.errstate(all="ignore"):
                want, _ = ora.render_streams(spheres, planes, cam, w, h, 1 << 16, spp, start)
            # both forms: the per-pixel chain, and the stream form (start-hit list, lanes that refill; every other time with
            # the sample chains cut into ordered passes of one or two samples)
            for form, batch in ((pkg.binding.FORM_AUTO, 0), (pkg.binding.FORM_STREAM, int(r.choice([0, 1, 2])))):
                ctx.set_option(pkg.binding.OPT_STREAMS_FORM, form)
                ctx.set_option(pkg.binding.OPT_STREAM_BATCH, batch)
                try:
                    ctx.upload_state(*start)
                    ctx.render(cam, limit, spp, pkg.STREAMS)
                    got = ctx.download_state()
                finally:
                    ctx.set_option(pkg.binding.OPT_STREAMS_FORM, pkg.binding.FORM_AUTO)
                    ctx.set_option(pkg.binding.OPT_STREAM_BATCH, 0)
                assert_planes_equal_up_to_nan_payload(got, want, what + (" streams, stream form, items of %d" % batch if form else " streams"))
        if case % 500 == 499:
            print("extreme fuzz: %d cases" % (case + 1), flush=True)


N_GLASS = int(os.environ.get("PTMI_FUZZ_GLASS_CASES", "150"))


def test_random_glass_scenes_tree_walk_and_stream_form(ctx, pkg, ora):
    """Random scenes with GLASS primitives (the build-defined extension; spec = the oracle): the per-pixel tree walk --
    start record, lane stack, sample chunks -- must equal the oracle's tree order BIT FOR BIT; every fourth case also runs
    the stream form, which must agree with the oracle's stream order to rounding with exact RNG planes.  A step cap keeps
    facing glass surfaces from bouncing a ray 65 536 times; the rays it cuts must be counted alike."""
    B = pkg.binding
    r = np.random.default_rng(SEED + 2)
    for case in range(N_GLASS):
        spheres, planes, cam, w, h, limit, spp = random_case(pkg, r)
        n_glass = 0
        for arr in (spheres, planes):
            for i in range(len(arr)):
                if r.random() < 0.45:
                    arr["brdf_tag"][i] = pkg.GLASS
                    arr["brdf_param"][i] = float(r.choice([1.0, 1.33, 1.5, 2.4, 0.7]))
                    arr["color"][i] = r.uniform(0.2, 1.0, 3)
                    arr["illuminance"][i] = float(r.choice([0.0, 0.0, 0.0, 3.0]))
                    n_glass += 1
        if case % 4 == 0:
            # The cases that also go through the stream form.  Its float atomics add a pixel's contributions in yet
            # another order, and contributions of opposite sign cancel (case 44 of this generator: terms of +-3e5 leaving
            # -8.25), which no relative tolerance survives.  Keep every contribution non-negative here: Glossy instead of
            # Matte (Matte's brdf is not clamped, Trace.hs:411), non-negative colours, unit plane normals (Schlick's
            # weight leaves [0, 1] otherwise).
            for arr in (spheres, planes):
                arr["brdf_tag"][arr["brdf_tag"] == pkg.MATTE] = pkg.GLOSSY
                arr["color"][:] = np.abs(arr["color"])
            for j in range(len(planes)):
                n = planes["direction"][j].astype(np.float64)
                planes["direction"][j] = (n / np.linalg.norm(n)).astype(np.float32)
        cap = int(r.choice([1, 2, 3, 4, 6, 9]))
        chunks = int(r.choice([0, 0, 1, 2, 3]))
        start = initial_planes(ora, w, h, seed0=int(r.integers(0, 2 ** 63)))
        if case % 7 == 3:                                          # a deep tree now and then: the lane stack overflows and must say so
            cap, w, h, spp = 24, min(w, 24), min(h, 12), 1
            start = [p[:h, :w].copy() for p in start]
        ctx.set_scene(spheres, planes)
        ctx.set_option(B.OPT_STREAM_STEP_CAP, cap)
        ctx.set_option(B.OPT_SPP_CHUNKS, chunks)
        try:
            ctx.resize(w, h)
            ctx.upload_state(*start)
            ctx.reset_stats()
            ctx.render(cam, limit, spp, pkg.STREAMS)
            got, st = ctx.download_state(), ctx.stats()
            with np.errstate(all="ignore"):
                want, live, dropped, longest, cut = ora.render_streams_tree(spheres, planes, cam, w, h, cap, spp, start)
            what = "glass case %d (%dx%d, %d+%d prims, %d glass, cap %d, spp %d, chunks %d)" % (case, w, h, len(spheres), len(planes), n_glass, cap, spp, chunks)
            if n_glass:
                assert_planes_equal(got, want, what + " tree walk")
                assert st["live_bounces"] == live and st["stream_rays_truncated"] == cut and st["stream_rays_dropped"] == dropped, what
            if case % 4 == 0 and n_glass:
                ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
                ctx.set_option(B.OPT_STREAM_CAPACITY, 64)
                ctx.set_option(B.OPT_STREAM_BATCH, int(r.choice([0, 0, 1, 2, 5])))    # samples per item: several passes over the start hits
                ctx.set_option(B.OPT_STREAM_GRADED, int(r.choice([1, 1, 0])))          # graded passes (the default) or round 3's uniform ones
                ctx.set_option(B.OPT_GLASS_BATCH, int(r.choice([0, 0, 2, 8, 64])))     # GLASS hits parked in their lanes (the default: off)
                ctx.set_option(B.OPT_STREAM_PASS_GROUPS, int(r.choice([0, 0, 1, 2, 3, 102, 103, 164])))   # the ticket order: groups of equal-size passes (default), pass by pass, fixed groups
                ctx.upload_state(*start)
                ctx.reset_stats()
                ctx.render(cam, limit, spp, pkg.STREAMS)
                got_s, st_s = ctx.download_state(), ctx.stats()
                for a, b in zip(got_s[3:], want[3:]):
                    assert np.array_equal(a, b), what
                if dropped:
                    # the tree walk dropped children (a lane holds sixteen waiting ones) and with them their subtrees; the stream form
                    # holds 64 rays per pixel and drops none of these: its rays and colours are the oracle's STREAM order's
                    with np.errstate(all="ignore"):
                        want, live, dropped_s, _, cut = ora.render_streams_wavefront(spheres, planes, cam, w, h, cap, spp, start, capacity_factor=64,
                                                                                      seed_rule=ora.SEED_KEEP_ACCUMULATOR, want_truncated=True)
                    assert dropped_s == 0, what
                assert st_s["live_bounces"] == live and st_s["stream_rays_truncated"] == cut and st_s["stream_rays_dropped"] == 0, what
                for a, b in zip(got_s[:3], want[:3]):                            # another order of the same non-negative terms
                    ratio = np.abs(a - b) / np.maximum(np.abs(b), 1e-3 + 1e-4 * np.max(np.abs(b)))
                    assert np.max(ratio) <= 1e-4, "%s: stream form vs tree order, worst %.3g at %d: %r vs %r" % (
                        what, float(np.max(ratio)), int(np.argmax(ratio)), a.reshape(-1)[np.argmax(ratio)], b.reshape(-1)[np.argmax(ratio)])
        finally:
            ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
            ctx.set_option(B.OPT_STREAM_BATCH, 0)
            ctx.set_option(B.OPT_STREAM_CAPACITY, 4)
            ctx.set_option(B.OPT_STREAM_STEP_CAP, 1 << 16)
            ctx.set_option(B.OPT_SPP_CHUNKS, 0)
            ctx.set_option(B.OPT_STREAM_GRADED, 1)
            ctx.set_option(B.OPT_GLASS_BATCH, 0)
            ctx.set_option(B.OPT_STREAM_PASS_GROUPS, 0)
