"""The reference's 8 intersection properties (test/Scene/Intersection/Tests.hs:32-121) evaluated on
the DEVICE through ptmi_eval_distance_to_{sphere,plane}, the way the reference evaluates single
expressions through its backend (Tests.hs:122-123) -- and bit-compared with the CPU oracle."""
import numpy as np
import pytest

import refprops as rp

pytestmark = pytest.mark.gpu
F = np.float32


def _rays(cases):
    return np.array([list(c["origin"]) + list(c["direction"]) for c in cases], F)


def eval_spheres(ctx, pkg, cases):
    sph = np.array([rp.make_sphere(pkg.world.SPHERE_DTYPE, c["pos"], c["radius"]) for c in cases])
    return sph, ctx.eval_distance_to_sphere(sph, _rays(cases))


def eval_planes(ctx, pkg, cases):
    pl = np.array([rp.make_plane(pkg.world.PLANE_DTYPE, c["pos"], c["nor"]) for c in cases])
    return pl, ctx.eval_distance_to_plane(pl, _rays(cases))


def test_sphere_intersection_position(ctx, pkg):
    cases = list(rp.sphere_intersection_cases())
    _, (just, t, nrm) = eval_spheres(ctx, pkg, cases)
    assert np.all(just == 1)
    for c, hp in zip(cases, nrm[:, :3]):
        assert tuple(rp.round_to(3, v) for v in hp) == c["expect_hit_pos_3dp"], c


def test_sphere_distance(ctx, pkg):
    cases = list(rp.sphere_distance_cases())
    _, (just, t, _) = eval_spheres(ctx, pkg, cases)
    assert np.all(just == 1)
    for c, tt in zip(cases, t):
        assert rp.round_to(1, tt) == c["expect_t_1dp"], (c, tt)


@pytest.mark.parametrize("gen", [rp.sphere_backface_cases, rp.sphere_backwards_cases])
def test_sphere_nothing(ctx, pkg, gen):
    _, (just, _, _) = eval_spheres(ctx, pkg, list(gen()))
    assert np.all(just == 0)


@pytest.mark.parametrize("gen", [rp.plane_straight_cases, rp.plane_angle_cases])
def test_plane_continuous(ctx, pkg, gen):
    cases = list(gen())
    _, (just, t, _) = eval_planes(ctx, pkg, cases)
    for c, j, tt in zip(cases, just, t):
        assert bool(j) == (c["expect"] is not None), c
        if j:
            assert tt == c["expect"], (c, tt)          # exact, as the reference's `===`


@pytest.mark.parametrize("gen", [rp.plane_straight_backface_cases, rp.plane_angle_backface_cases])
def test_plane_backface(ctx, pkg, gen):
    _, (just, _, _) = eval_planes(ctx, pkg, list(gen()))
    assert np.all(just == 0)


def test_device_equals_oracle_bitwise_on_random_rays(ctx, pkg, ora):
    """distanceTo + hit for random rays/primitives: device == oracle, bit for bit (incl. Nothing)."""
    r = np.random.default_rng(99)
    n = 3000
    o = r.uniform(-20, 20, (n, 3)).astype(F)
    d = r.normal(0, 1, (n, 3)).astype(F)
    d = np.array([rp.l_normalize(v) for v in d], F)
    rays = np.concatenate([o, d], 1)
    sph = np.array([rp.make_sphere(pkg.world.SPHERE_DTYPE, p, rad) for p, rad in
                    zip(r.uniform(-20, 20, (n, 3)).astype(F), r.uniform(0.1, 15, n).astype(F))])
    just, t, nrm = ctx.eval_distance_to_sphere(sph, rays)
    hits = 0
    for i in range(n):
        want = ora.distance_to_sphere(o[i], d[i], sph[i])
        assert bool(just[i]) == (want is not None)
        if want is not None:
            hits += 1
            assert t[i].view(np.uint32) == want.view(np.uint32)
            pos, nor, _ = ora.hit_sphere(o[i], d[i], want, sph[i])
            assert np.array_equal(nrm[i, :3].view(np.uint32), pos.view(np.uint32))
            assert np.array_equal(nrm[i, 3:].view(np.uint32), nor.view(np.uint32))
    assert hits > 50
    pl = np.array([rp.make_plane(pkg.world.PLANE_DTYPE, p, nn) for p, nn in
                   zip(r.uniform(-20, 20, (n, 3)).astype(F), r.normal(0, 1, (n, 3)).astype(F))])
    just, t, nrm = ctx.eval_distance_to_plane(pl, rays)
    for i in range(n):
        want = ora.distance_to_plane(o[i], d[i], pl[i])
        assert bool(just[i]) == (want is not None)
        if want is not None:
            assert t[i].view(np.uint32) == want.view(np.uint32)
            assert np.array_equal(nrm[i, 3:], pl[i]["direction"])


def test_sphere_normals_equal_oracle_when_every_lane_hits(ctx, pkg, ora):
    """hit's normalisation divides three components by one length; the device shares the reciprocal when no operand of any
    lane needs scaling (div3_by_length) and takes the compiler's division otherwise.  Rays aimed AT their spheres, so that
    whole waves hit: ordinary magnitudes (the shared form), then the same with axis-aligned rays (zero components), tiny
    and huge spheres (the fallback).  Bit for bit against the oracle's IEEE divisions."""
    r = np.random.default_rng(2024)

    def batch(n, radius, aligned):
        rad = radius(n).astype(F)
        c = (r.uniform(-10, 10, (n, 3)) * rad[:, None]).astype(F)
        if aligned:
            axis = r.integers(0, 3, n)
            off = np.zeros((n, 3), F)
            off[np.arange(n), axis] = (rad * F(3.0)).astype(F)
            o = (c + off).astype(F)
            d = np.zeros((n, 3), F)
            d[np.arange(n), axis] = F(-1.0)
        else:
            o = (c + (r.normal(0, 1, (n, 3)) * rad[:, None] * 4).astype(F)).astype(F)
            target = (c + (r.uniform(-0.5, 0.5, (n, 3)) * rad[:, None]).astype(F)).astype(F)
            v = (target - o).astype(np.float64)
            d = (v / np.linalg.norm(v, axis=1)[:, None]).astype(F)       # (linear's normalize leaves tiny vectors unscaled)
        sph = np.array([rp.make_sphere(pkg.world.SPHERE_DTYPE, p, q) for p, q in zip(c, rad)])
        just, t, nrm = ctx.eval_distance_to_sphere(sph, np.concatenate([o, d], 1))
        hits = 0
        for i in range(n):
            want = ora.distance_to_sphere(o[i], d[i], sph[i])
            assert bool(just[i]) == (want is not None)
            if want is not None:
                hits += 1
                pos, nor, _ = ora.hit_sphere(o[i], d[i], want, sph[i])
                assert np.array_equal(nrm[i, :3].view(np.uint32), pos.view(np.uint32))
                assert np.array_equal(nrm[i, 3:].view(np.uint32), nor.view(np.uint32)), (i, nrm[i, 3:], nor)
        return hits

    n = 4096
    assert batch(n, lambda k: np.exp(r.uniform(np.log(0.05), np.log(500.0), k)), False) > 0.9 * n
    assert batch(n, lambda k: r.uniform(0.5, 3.0, k), True) > 0.9 * n
    assert batch(n, lambda k: np.exp(r.uniform(np.log(1e-7), np.log(1e-5), k)), False) > 0
    assert batch(n, lambda k: np.exp(r.uniform(np.log(1e13), np.log(1e17), k)), False) > 0

    # tiny components of a long vector: the quotients approach (shared form, radius <= 2^20) or enter (compiler's form) the
    # denormal range, where an unscaled division sequence would round differently
    for radius in (5.0e5, 1.0e8, 3.0e11):
        k = 2048
        tiny = np.exp(r.uniform(np.log(1e-30), np.log(1e-24), (k, 2))).astype(F) * r.choice([-1.0, 1.0], (k, 2)).astype(F)
        o = np.concatenate([tiny, np.full((k, 1), 2.0 * radius, F)], 1).astype(F)
        d = np.tile(np.array([0.0, 0.0, -1.0], F), (k, 1))
        sph = np.array([rp.make_sphere(pkg.world.SPHERE_DTYPE, np.zeros(3, F), F(radius)) for _ in range(k)])
        just, t, nrm = ctx.eval_distance_to_sphere(sph, np.concatenate([o, d], 1))
        assert np.all(just == 1)
        for i in range(k):
            want = ora.distance_to_sphere(o[i], d[i], sph[i])
            pos, nor, _ = ora.hit_sphere(o[i], d[i], want, sph[i])
            assert np.array_equal(nrm[i, 3:].view(np.uint32), nor.view(np.uint32)), (radius, i, nrm[i, 3:], nor)


def test_sqrt_rn_around_its_2_to_the_minus_96_boundary(ctx, pkg, ora):
    """sqrt_rn (v_sqrt_f32 + two FMA residuals) is exact for x = 0 or x >= 2^-96; below that the whole wave takes the compiler's
    scaled square root.  Spheres with radii around 2^-48 met head-on from 2^-40 ... 2^-45 away (d2 = 0 exactly, so the argument
    of the square root is rad^2, on both sides of 2^-96, and t = tca - sqrt(rad^2) shows its every bit): whole waves on the fast
    path, whole waves on the scaled path, and waves that mix both.  Bit for bit against the oracle's sqrtf."""
    r = np.random.default_rng(96)
    boundary = 2.0 ** -48                                    # rad^2 == 2^-96

    def batch(radii):
        n = radii.size
        k = (2.0 ** -r.integers(40, 46, n)).astype(F)
        c = np.zeros((n, 3), F); c[:, 2] = k
        o = np.zeros((n, 3), F)
        d = np.tile(np.array([0.0, 0.0, 1.0], F), (n, 1))
        sph = np.array([rp.make_sphere(pkg.world.SPHERE_DTYPE, p, q) for p, q in zip(c, radii.astype(F))])
        just, t, _ = ctx.eval_distance_to_sphere(sph, np.concatenate([o, d], 1))
        for i in range(n):
            want = ora.distance_to_sphere(o[i], d[i], sph[i])
            assert bool(just[i]) == (want is not None), i
            if want is not None:
                assert np.float32(t[i]).view(np.uint32) == np.float32(want).view(np.uint32), (i, float(radii[i]), float(t[i]), float(want))
        return int(just.sum())

    n = 2048
    above = np.exp(r.uniform(np.log(boundary * 1.001), np.log(boundary * 4), n))     # (rad < 2^-45 <= the distance: every ray hits)
    below = np.exp(r.uniform(np.log(boundary / 4096), np.log(boundary * 0.999), n))
    assert batch(above) == n                                 # every lane on the fast path
    assert batch(below) == n                                 # every lane below 2^-96: the scaled path
    mixed = np.where(r.random(n) < 0.5, above, below)        # both in one wave: the wave takes the scaled path
    assert batch(mixed) == n
    edge = boundary * (1.0 + r.integers(-3, 4, n) * 2.0 ** -23)     # the neighbours of 2^-48 themselves
    assert batch(edge) == n


def test_device_sincos_equals_oracle_and_libm(ctx, ora):
    """The device's sin/cos (binary64 evaluation of glibc's algorithm) == oracle == host libm, bitwise."""
    r = np.random.default_rng(5)
    x = np.concatenate([
        r.uniform(-np.pi / 2, np.pi / 2, 200000), r.uniform(-130, 130, 50000), r.normal(0, 1e5, 20000),
        10.0 ** r.uniform(-38, 38, 20000), -(10.0 ** r.uniform(-38, 38, 20000)),
        [0.0, -0.0, 0.75, 0.7499999, 2.0 ** -12, 119.99999, 120.0, 1e30, 3.4e38, np.inf, -np.inf, np.nan],
    ]).astype(F)
    s, c = ctx.eval_sincos(x)
    sub = r.choice(x.size - 3, 20000, replace=False)
    so, co = ora.sincos_array(x[sub])
    assert np.array_equal(s[sub].view(np.uint32), so.view(np.uint32))
    assert np.array_equal(c[sub].view(np.uint32), co.view(np.uint32))
    with np.errstate(invalid="ignore"):
        # numpy's float32 sin/cos are not glibc's; compare through the oracle only, plus sanity vs float64
        assert np.nanmax(np.abs(s[:-3] - np.sin(x[:-3].astype(np.float64)))) < 1e-6
        assert np.nanmax(np.abs(c[:-3] - np.cos(x[:-3].astype(np.float64)))) < 1e-6
    assert np.all(np.isnan(s[-3:])) and np.all(np.isnan(c[-3:]))
