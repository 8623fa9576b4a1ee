"""Pins the CPU oracle against the reference's own known-answer tests for this path: the 8
Hedgehog properties of test/Scene/Intersection/Tests.hs:32-121 (the only tests the reference has)."""
import numpy as np

import refprops as rp

F = np.float32


def _sphere_t(ora, c):
    s = rp.make_sphere(ora.SPHERE_DTYPE, c["pos"], c["radius"])
    return s, ora.distance_to_sphere(c["origin"], c["direction"], s)


def _plane_t(ora, c):
    p = rp.make_plane(ora.PLANE_DTYPE, c["pos"], c["nor"])
    return ora.distance_to_plane(c["origin"], c["direction"], p)


def test_sphere_intersection_position(ora):
    for c in rp.sphere_intersection_cases():
        s, t = _sphere_t(ora, c)
        assert t is not None                                   # fromJust
        pos, _nor, _mat = ora.hit_sphere(c["origin"], c["direction"], t, s)
        got = tuple(rp.round_to(3, v) for v in pos)
        assert got == c["expect_hit_pos_3dp"], (c, got)


def test_sphere_distance(ora):
    for c in rp.sphere_distance_cases():
        _, t = _sphere_t(ora, c)
        assert t is not None and rp.round_to(1, t) == c["expect_t_1dp"], (c, t)


def test_sphere_backface_culling(ora):
    for c in rp.sphere_backface_cases():
        assert _sphere_t(ora, c)[1] is None, c


def test_sphere_no_backwards_intersections(ora):
    for c in rp.sphere_backwards_cases():
        assert _sphere_t(ora, c)[1] is None, c


def test_plane_continuous_straight_on(ora):
    for c in rp.plane_straight_cases():
        t = _plane_t(ora, c)
        assert (t is None) == (c["expect"] is None) and (t is None or t == c["expect"]), (c, t)


def test_plane_backface_straight_on(ora):
    for c in rp.plane_straight_backface_cases():
        assert _plane_t(ora, c) is None, c


def test_plane_continuous_angles(ora):
    for c in rp.plane_angle_cases():
        t = _plane_t(ora, c)
        assert t is not None and c["expect"] is not None and t == c["expect"], (c, t)    # exact, as `===`


def test_plane_backface_angles(ora):
    for c in rp.plane_angle_backface_cases():
        assert _plane_t(ora, c) is None, c


def test_hit_returns_material_and_plane_normal_as_stored(ora):
    # Intersection.hs:29-32, :64 -- the plane normal is neither normalised nor flipped
    p = rp.make_plane(ora.PLANE_DTYPE, (0, 0, 5), (0, 0, -2.5))
    pos, nor, mat = ora.hit_plane((0, 0, 0), (0, 0, 1), 5.0, p)
    assert tuple(pos) == (0, 0, 5) and tuple(nor) == (0, 0, -2.5)
    assert tuple(mat[0]) == (1, 1, 1) and mat[1] == 1 and mat[2] == 0 and mat[3] == 1
