"""The reference's own 8 Hedgehog properties (test/Scene/Intersection/Tests.hs:32-121) restated as
seeded case generators + expected values, so the SAME cases can be put to the CPU oracle
(tests/test_oracle_intersection.py) and to the device through the C ABI
(tests/test_gpu_intersection.py).  Host-side `linear` arithmetic used by the reference's test
code (L.normalize, L.dot, roundTo) is restated here in numpy float32, independently of both."""
import numpy as np

F = np.float32
N_CASES = 400          # Hedgehog's default is 100 per property


def rng(tag):
    return np.random.default_rng([0x5EED, tag])


def l_normalize(v):
    """linear: normalize v = if nearZero l || nearZero (1-l) then v else fmap (/ sqrt l) v"""
    v = np.asarray(v, F)
    l = F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2])
    l = F(l)
    if abs(l) <= F(1e-6) or abs(F(F(1.0) - l)) <= F(1e-6):
        return v
    s = np.sqrt(l, dtype=F)
    return np.array([v[0] / s, v[1] / s, v[2] / s], F)


def round_to(places, x):
    """Tests.hs:146-147: fromInteger (round $ i * (10 ^ places)) / (10.0 ^^ places); `round` is half-even."""
    scale = F(10 ** places)
    return F(np.rint(F(F(x) * scale)) / scale)


def dummy_material():
    """Tests.hs:139-143"""
    return dict(color=(1.0, 1.0, 1.0), illuminance=1.0, brdf_tag=0, brdf_param=1.0)


def make_sphere(dtype, pos, radius):
    s = np.zeros((), dtype)
    s["position"] = pos; s["radius"] = radius
    m = dummy_material()
    s["color"] = m["color"]; s["illuminance"] = m["illuminance"]; s["brdf_tag"] = m["brdf_tag"]; s["brdf_param"] = m["brdf_param"]
    return s


def make_plane(dtype, pos, nor):
    p = np.zeros((), dtype)
    p["position"] = pos; p["direction"] = nor
    m = dummy_material()
    p["color"] = m["color"]; p["illuminance"] = m["illuminance"]; p["brdf_tag"] = m["brdf_tag"]; p["brdf_param"] = m["brdf_param"]
    return p


# Each generator yields dicts: origin, direction, prim-args, and the expectation.
def sphere_intersection_cases():
    """Tests.hs:35-42  intersection ((x, 0, x), x) = (0, 0, x)"""
    ds = np.concatenate([[F(0.0), F(100.0)], rng(1).uniform(0.0, 100.0, N_CASES).astype(F)])
    for d in ds:
        yield dict(origin=(0, 0, 0), direction=(0, 0, 1), pos=(d, 0.0, d), radius=d,
                   expect_hit_pos_3dp=(F(0.0), F(0.0), round_to(3, d)))


def sphere_distance_cases():
    """Tests.hs:43-58  distanceTo ((x, x, x), y) = ||y|| - y + ||(x - y)||"""
    r = rng(2)
    direction = l_normalize((1.0, 1.0, 1.0))       # Accelerate normalize of a constant: same definition
    for d, off in zip(r.uniform(0.1, 100.0, N_CASES).astype(F), r.uniform(0.1, 100.0, N_CASES).astype(F)):
        pos = F(d + off)
        expected = F(F(np.sqrt(F(F(3) * F(d * d)), dtype=F) - d) + np.sqrt(F(F(3) * F(off * off)), dtype=F))
        yield dict(origin=(0, 0, 0), direction=tuple(direction), pos=(pos, pos, pos), radius=d,
                   expect_t_1dp=round_to(1, expected))


def sphere_backface_cases():
    """Tests.hs:59-66"""
    r = rng(3)
    for d, v in zip(r.uniform(0.1, 100.0, N_CASES).astype(F), r.uniform(-1.0, 1.0, (N_CASES, 3)).astype(F)):
        yield dict(origin=(0, 0, 0), direction=tuple(l_normalize(v)), pos=(0, 0, 0), radius=d, expect=None)


def sphere_backwards_cases():
    """Tests.hs:67-72"""
    for v in rng(4).uniform(-1.0, 1.0, (N_CASES, 3)).astype(F):
        d = l_normalize(v)
        yield dict(origin=(0, 0, 0), direction=tuple(d), pos=tuple(-d), radius=F(0.1), expect=None)


def _points(tag):
    pts = rng(tag).uniform(-1000.0, 1000.0, (N_CASES, 3)).astype(F)
    pts[0] = (3.0, -4.0, 0.0)      # z = 0 boundary: `z >= 0` -> Just 0
    return pts


def plane_straight_cases():
    """Tests.hs:78-85"""
    for p in _points(5):
        yield dict(origin=(0, 0, 0), direction=(0, 0, 1), pos=tuple(p), nor=(0, 0, -1),
                   expect=(p[2] if p[2] >= 0 else None))


def plane_straight_backface_cases():
    """Tests.hs:86-93"""
    for p in _points(6):
        yield dict(origin=(0, 0, 0), direction=(0, 0, 1), pos=tuple(p), nor=(0, 0, 1), expect=None)


def plane_angle_cases():
    """Tests.hs:94-107 (exact === on Float)"""
    for x, y in rng(7).uniform(-1000.0, 1000.0, (N_CASES, 2)).astype(F):
        d = l_normalize((x, y, 1.0))
        cos_angle = F(F(F(d[0] * F(0)) + F(d[1] * F(0))) + F(d[2] * F(1)))     # L.dot dir (V3 0 0 1)
        dist = F(F(1.0) / cos_angle)
        yield dict(origin=(0, 0, 0), direction=tuple(d), pos=(0, 0, 1), nor=(0, 0, -1),
                   expect=(dist if dist >= 0 else None))


def plane_angle_backface_cases():
    """Tests.hs:108-115"""
    for x, y in rng(8).uniform(-1000.0, 1000.0, (N_CASES, 2)).astype(F):
        d = l_normalize((x, y, 1.0))
        yield dict(origin=(0, 0, 0), direction=tuple(d), pos=(0, 0, 1), nor=(0, 0, 1), expect=None)
