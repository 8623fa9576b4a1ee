"""CPU ORACLE -- test infrastructure only (see pt_oracle.h for the parity status).

ctypes wrapper of oracle/libptoracle.so, the plain-C restatement of the reference's render path.
May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY; the
product (haskell-path-tracer_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("PTMI_ORACLE_LIB") or os.path.join(HERE, "libptoracle.so")   # (the override: tests/test_oracle_sanitized.py loads an ASan / UBSan build)

# independent declarations of the flat records (must equal the byte layout of pt_oracle.h)
SPHERE_DTYPE = np.dtype([("position", "<f4", 3), ("radius", "<f4"), ("color", "<f4", 3),
                         ("illuminance", "<f4"), ("brdf_tag", "<i4"), ("brdf_param", "<f4")])
PLANE_DTYPE = np.dtype([("position", "<f4", 3), ("direction", "<f4", 3), ("color", "<f4", 3),
                        ("illuminance", "<f4"), ("brdf_tag", "<i4"), ("brdf_param", "<f4")])
CAMERA_DTYPE = np.dtype([("position", "<f4", 3), ("rotation", "<f4", 3), ("fov", "<i8")])


class _V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class _Ray(C.Structure):
    _fields_ = [("origin", _V3), ("direction", _V3)]


class _MaybeFloat(C.Structure):
    _fields_ = [("is_just", C.c_int), ("value", C.c_float)]


class _Material(C.Structure):
    _fields_ = [("color", _V3), ("illuminance", C.c_float), ("brdf_tag", C.c_int32), ("brdf_param", C.c_float)]


class _MaybeHit(C.Structure):
    _fields_ = [("is_just", C.c_int), ("normal_p", _Ray), ("material", _Material)]


class _Scene(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("n_spheres", C.c_int), ("planes", C.c_void_p), ("n_planes", C.c_int)]


class _Sfc(C.Structure):
    _fields_ = [("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32), ("counter", C.c_uint32)]


class _Opts(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("n_rows", C.c_int), ("streams_seed_rule", C.c_int), ("n_threads", C.c_int)]


SEED_KEEP_ACCUMULATOR, SEED_FROM_RESULT = 0, 1        # ORA_SEED_* (assumption A5, pt_oracle.h)
GLASS_TAG = 2


def default_seed_rule(spheres, planes):
    """What libptmi's PTMI_SEED_AUTO resolves to: `combine new old` (the result's seed) unless a primitive splits rays."""
    tags = list(np.asarray(spheres)["brdf_tag"].reshape(-1)) + list(np.asarray(planes)["brdf_tag"].reshape(-1))
    return SEED_KEEP_ACCUMULATOR if any(int(t) == GLASS_TAG for t in tags) else SEED_FROM_RESULT


def build(force=False):
    if os.environ.get("PTMI_ORACLE_LIB"):
        return LIB
    src = [os.path.join(HERE, f) for f in ("pt_oracle.c", "pt_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.run(["make", "-s", "-C", HERE, "-f", os.path.join(HERE, "Makefile")], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.ora_sinf.restype = C.c_float; L.ora_sinf.argtypes = [C.c_float]
        L.ora_cosf.restype = C.c_float; L.ora_cosf.argtypes = [C.c_float]
        L.ora_distance_to_sphere.restype = _MaybeFloat; L.ora_distance_to_sphere.argtypes = [_Ray, C.c_void_p]
        L.ora_distance_to_plane.restype = _MaybeFloat; L.ora_distance_to_plane.argtypes = [_Ray, C.c_void_p]
        L.ora_hit_sphere.restype = _MaybeHit; L.ora_hit_sphere.argtypes = [_Ray, C.c_float, C.c_void_p]
        L.ora_hit_plane.restype = _MaybeHit; L.ora_hit_plane.argtypes = [_Ray, C.c_float, C.c_void_p]
        L.ora_check_hit.restype = _MaybeHit; L.ora_check_hit.argtypes = [C.POINTER(_Scene), _Ray]
        L.ora_render_inline.restype = C.c_int64
        L.ora_render_inline.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p] + [C.c_void_p] * 7 + [C.c_int]
        L.ora_render_inline_ex.restype = C.c_int64
        L.ora_render_inline_ex.argtypes = L.ora_render_inline.argtypes + [C.POINTER(_Opts)]
        L.ora_render_streams.restype = C.c_int64
        L.ora_render_streams.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.ora_render_streams_wavefront.restype = C.c_int64
        L.ora_render_streams_wavefront.argtypes = ([C.POINTER(_Scene), C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 7 +
                                                   [C.POINTER(C.c_int64), C.POINTER(C.c_int)])
        L.ora_render_streams_ex.restype = C.c_int64
        L.ora_render_streams_ex.argtypes = L.ora_render_streams.argtypes + [C.POINTER(_Opts), C.POINTER(C.c_int64)]
        L.ora_render_streams_wavefront_ex.restype = C.c_int64
        L.ora_render_streams_wavefront_ex.argtypes = (L.ora_render_streams_wavefront.argtypes +
                                                      [C.POINTER(_Opts), C.POINTER(C.c_int64)])
        L.ora_render_streams_tree.restype = C.c_int64
        L.ora_render_streams_tree.argtypes = L.ora_render_streams_wavefront_ex.argtypes
        L.ora_gen_seeds.restype = None
        L.ora_gen_seeds.argtypes = [C.c_uint64, C.c_int64, C.c_int64] + [C.c_void_p] * 4
        L.ora_sfc32_next.restype = C.c_uint32; L.ora_sfc32_next.argtypes = [C.POINTER(_Sfc)]
        L.ora_random_float.restype = C.c_float; L.ora_random_float.argtypes = [C.POINTER(_Sfc)]
        L.ora_sfc32_seed3.restype = _Sfc; L.ora_sfc32_seed3.argtypes = [C.c_uint32] * 3
        L.ora_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _ray(o, d):
    return _Ray(_V3(*[float(np.float32(v)) for v in o]), _V3(*[float(np.float32(v)) for v in d]))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def sinf(x):
    return np.float32(lib().ora_sinf(np.float32(x)))


def cosf(x):
    return np.float32(lib().ora_cosf(np.float32(x)))


def sincos_array(x):
    x = np.ascontiguousarray(x, np.float32)
    L = lib()
    s = np.array([L.ora_sinf(v) for v in x.tolist()], np.float32)
    c = np.array([L.ora_cosf(v) for v in x.tolist()], np.float32)
    return s, c


def distance_to_sphere(origin, direction, sphere):
    """-> None | float32   (Intersection.hs:39-48)"""
    s = np.ascontiguousarray(sphere, SPHERE_DTYPE)
    m = lib().ora_distance_to_sphere(_ray(origin, direction), _p(s))
    return np.float32(m.value) if m.is_just else None


def distance_to_plane(origin, direction, plane):
    p = np.ascontiguousarray(plane, PLANE_DTYPE)
    m = lib().ora_distance_to_plane(_ray(origin, direction), _p(p))
    return np.float32(m.value) if m.is_just else None


def _hit_tuple(h):
    if not h.is_just:
        return None
    pos = np.array([h.normal_p.origin.x, h.normal_p.origin.y, h.normal_p.origin.z], np.float32)
    nor = np.array([h.normal_p.direction.x, h.normal_p.direction.y, h.normal_p.direction.z], np.float32)
    mat = (np.array([h.material.color.x, h.material.color.y, h.material.color.z], np.float32),
           np.float32(h.material.illuminance), int(h.material.brdf_tag), np.float32(h.material.brdf_param))
    return pos, nor, mat


def hit_sphere(origin, direction, t, sphere):
    s = np.ascontiguousarray(sphere, SPHERE_DTYPE)
    return _hit_tuple(lib().ora_hit_sphere(_ray(origin, direction), np.float32(t), _p(s)))


def hit_plane(origin, direction, t, plane):
    p = np.ascontiguousarray(plane, PLANE_DTYPE)
    return _hit_tuple(lib().ora_hit_plane(_ray(origin, direction), np.float32(t), _p(p)))


def _scene(spheres, planes):
    s = np.ascontiguousarray(spheres, SPHERE_DTYPE)
    p = np.ascontiguousarray(planes, PLANE_DTYPE)
    sc = _Scene(s.ctypes.data if s.size else None, s.size, p.ctypes.data if p.size else None, p.size)
    return sc, (s, p)


def check_hit(spheres, planes, origin, direction):
    sc, keep = _scene(spheres, planes)
    return _hit_tuple(lib().ora_check_hit(C.byref(sc), _ray(origin, direction)))


def gen_seeds(seed0, first_index, n):
    out = [np.empty(n, np.uint32) for _ in range(4)]
    lib().ora_gen_seeds(C.c_uint64(seed0), first_index, n, *[_p(a) for a in out])
    return tuple(out)


def sfc32_stream(state, n):
    """n raw outputs and n floats from two copies of `state` (a, b, c, counter)."""
    L = lib()
    s1, s2 = _Sfc(*[int(v) for v in state]), _Sfc(*[int(v) for v in state])
    raw = np.array([L.ora_sfc32_next(C.byref(s1)) for _ in range(n)], np.uint32)
    flt = np.array([L.ora_random_float(C.byref(s2)) for _ in range(n)], np.float32)
    return raw, flt, (s1.a, s1.b, s1.c, s1.counter)


def sfc32_seed3(a, b, c):
    s = lib().ora_sfc32_seed3(int(a), int(b), int(c))
    return (s.a, s.b, s.c, s.counter)


def _opts(rows, seed_rule, n_threads=1):
    """-> (_Opts, keep-alive) for the *_ex functions; rows = image row of every held row (a partition) or None."""
    o = _Opts(None, 0, int(seed_rule), int(n_threads))
    keep = None
    if rows is not None:
        keep = np.ascontiguousarray(rows, np.int32)
        o.rows, o.n_rows = keep.ctypes.data, keep.size
    return o, keep


def _copies(planes_in, n_rows, width):
    return [np.array(a, dtype=np.float32, copy=True).reshape(n_rows, width) for a in planes_in[:3]] + \
           [np.array(a, dtype=np.uint32, copy=True).reshape(n_rows, width) for a in planes_in[3:]]


def render_inline(spheres, planes, camera, width, height, bounce_limit, n_spp, planes_in,
                  screen=None, n_threads=1, rows=None):
    """n_spp applications of `render Inline` to copies of the 7 planes; returns (planes_out, live_bounces).
    rows: the image rows the planes hold (default all) -- a row-stripe part of the width x height image."""
    sc, keep = _scene(spheres, planes)
    cam = np.ascontiguousarray(camera, CAMERA_DTYPE)
    o, keep_rows = _opts(rows, 0)
    outs = _copies(planes_in, height if rows is None else len(rows), width)
    sx = sy = None
    if screen is not None:
        sx = np.ascontiguousarray(screen[0], np.int64)
        sy = np.ascontiguousarray(screen[1], np.int64)
    live = lib().ora_render_inline_ex(C.byref(sc), _p(cam), width, height, bounce_limit, n_spp,
                                      _p(sx), _p(sy), *[_p(a) for a in outs], n_threads, C.byref(o))
    return tuple(outs), int(live)


def render_streams(spheres, planes, camera, width, height, max_iterations, n_spp, planes_in,
                   rows=None, seed_rule=None, want_truncated=False, n_threads=1):
    """seed_rule None = the library's default (default_seed_rule); n_threads > 1: OpenMP over rows (pixels are independent)."""
    sc, keep = _scene(spheres, planes)
    cam = np.ascontiguousarray(camera, CAMERA_DTYPE)
    o, keep_rows = _opts(rows, default_seed_rule(spheres, planes) if seed_rule is None else seed_rule, n_threads)
    outs = _copies(planes_in, height if rows is None else len(rows), width)
    truncated = C.c_int64(0)
    live = lib().ora_render_streams_ex(C.byref(sc), _p(cam), width, height, max_iterations, n_spp,
                                       *[_p(a) for a in outs], C.byref(o), C.byref(truncated))
    if want_truncated:
        return tuple(outs), int(live), int(truncated.value)
    return tuple(outs), int(live)


def render_streams_wavefront(spheres, planes, camera, width, height, hard_cap, n_spp, planes_in, capacity_factor=4,
                             rows=None, seed_rule=None, want_truncated=False):
    """Streams as a stream (supports the build-defined GLASS extension). -> (planes, live, dropped, steps[, truncated])
    seed_rule None = the library's default (default_seed_rule)."""
    sc, keep = _scene(spheres, planes)
    cam = np.ascontiguousarray(camera, CAMERA_DTYPE)
    o, keep_rows = _opts(rows, default_seed_rule(spheres, planes) if seed_rule is None else seed_rule)
    outs = _copies(planes_in, height if rows is None else len(rows), width)
    dropped, steps, truncated = C.c_int64(0), C.c_int(0), C.c_int64(0)
    live = lib().ora_render_streams_wavefront_ex(C.byref(sc), _p(cam), width, height, hard_cap, n_spp, capacity_factor,
                                                 *[_p(a) for a in outs], C.byref(dropped), C.byref(steps),
                                                 C.byref(o), C.byref(truncated))
    if want_truncated:
        return tuple(outs), int(live), int(dropped.value), int(steps.value), int(truncated.value)
    return tuple(outs), int(live), int(dropped.value), int(steps.value)


def render_streams_tree(spheres, planes, camera, width, height, hard_cap, n_spp, planes_in, stack_depth=16, rows=None, n_threads=1):
    """Streams with ray splitting visited per pixel, depth first (the device's tree-walk order); n_threads > 1: OpenMP over rows.
    -> (planes, live, dropped, longest_lineage, truncated)"""
    sc, keep = _scene(spheres, planes)
    cam = np.ascontiguousarray(camera, CAMERA_DTYPE)
    o, keep_rows = _opts(rows, 0, n_threads)
    outs = _copies(planes_in, height if rows is None else len(rows), width)
    dropped, longest, truncated = C.c_int64(0), C.c_int(0), C.c_int64(0)
    live = lib().ora_render_streams_tree(C.byref(sc), _p(cam), width, height, hard_cap, n_spp, stack_depth,
                                         *[_p(a) for a in outs], C.byref(dropped), C.byref(longest),
                                         C.byref(o), C.byref(truncated))
    return tuple(outs), int(live), int(dropped.value), int(longest.value), int(truncated.value)


def render_streams_wavefront_rows(spheres, planes, camera, width, height, hard_cap, n_spp, planes_in, rows, capacity_factor=8, n_threads=1):
    """ora_render_streams_wavefront over the image rows `rows` (planes_in: their 7 planes, [len(rows)][width]), one row per call and
    the calls on n_threads threads: the stream order of a pixel's additions does not depend on which other pixels share the stream,
    so this equals one call over all the rows -- and finishes in seconds where that takes minutes.  -> (planes, live, dropped)"""
    from concurrent.futures import ThreadPoolExecutor
    rows = [int(v) for v in rows]

    def one(k):
        start = [np.asarray(a)[k:k + 1] for a in planes_in]
        return render_streams_wavefront(spheres, planes, camera, width, height, hard_cap, n_spp, start, capacity_factor=capacity_factor, rows=[rows[k]])
    with ThreadPoolExecutor(max(1, int(n_threads))) as pool:
        res = list(pool.map(one, range(len(rows))))
    outs = tuple(np.concatenate([r[0][p] for r in res], axis=0) for p in range(7))
    return outs, sum(r[1] for r in res), sum(r[2] for r in res)


def max_threads():
    return int(lib().ora_max_threads())
