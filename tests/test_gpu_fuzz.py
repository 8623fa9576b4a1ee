"""Randomised parity campaign: random scenes (overlapping / tiny / huge / zero-radius spheres, un-normalised
plane normals, BRDF parameters outside their documented ranges, emissive everything), random cameras
(positions inside primitives, rotations beyond 120 rad, odd fields of view), random image shapes, bounce
limits and sample counts -- device vs oracle, BIT-EXACT on all seven planes, for both algorithms.  Seeded:
the same cases every run.  This is where a fast path that is only "almost always" equal to the literal
fold, square root or sin/cos would show."""
import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu
N_CASES = 600


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


def test_random_scenes_inline_and_streams(ctx, pkg, ora):
    r = np.random.default_rng(20260101)
    checked = 0
    for case in range(N_CASES):
        spheres, planes, cam, w, h, limit, spp = random_case(pkg, r)
        start = initial_planes(ora, w, h, seed0=int(r.integers(0, 2 ** 63)))
        ctx.set_scene(spheres, planes)
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.reset_stats()
        ctx.render(cam, limit, spp, pkg.INLINE)
        got = ctx.download_state()
        live_gpu = ctx.stats()["live_bounces"]
        with np.errstate(all="ignore"):
            want, live = ora.render_inline(spheres, planes, cam, w, h, limit, spp, start)
        assert_planes_equal(got, want, "fuzz case %d inline (%dx%d, %d+%d prims, limit %d, spp %d)" % (case, w, h, len(spheres), len(planes), limit, spp))
        assert live_gpu == live
        if case % 3 == 0:
            ctx.upload_state(*start)
            ctx.render(cam, limit, spp, pkg.STREAMS)
            got = ctx.download_state()
            with np.errstate(all="ignore"):
                want, _ = ora.render_streams(spheres, planes, cam, w, h, 1 << 16, spp, start)
            assert_planes_equal(got, want, "fuzz case %d streams" % case)
        checked += 1
    assert checked == N_CASES
