"""ctypes binding of include/ptmi.h -- the Python face of the C ABI used by tests and bench.py.

This is a thin mirror: one method per entry point, numpy arrays for host planes, optional
torch tensors for caller-owned device planes (torch is plumbing for device memory, streams and
torch.distributed only).  There is no fallback: if libptmi.so is missing or no GPU is present
the calls raise PtmiError.
"""
import ctypes as C
import os

import numpy as np

from . import _build
from .world import CAMERA_DTYPE, PLANE_DTYPE, SPHERE_DTYPE, INLINE, STREAMS

OPT_STREAMS_SEED_RULE, OPT_STREAM_STEP_CAP, OPT_STREAM_CAPACITY, OPT_STREAMS_FORM, OPT_STREAM_BATCH, OPT_SPP_CHUNKS, OPT_ARITHMETIC = 1, 2, 3, 4, 5, 6, 7
OPT_STREAM_TAIL, OPT_ORDERED_PASSES, OPT_GLASS_BATCH, OPT_STREAM_GRADED, OPT_SNAPSHOT_BUDGET_MB, OPT_STREAM_PASS_GROUPS, OPT_CHAIN_SLOTS, OPT_PASS_HANDOFF = 8, 9, 10, 11, 12, 13, 14, 15
CHAIN_CONSUME = 1
HANDOFF_FENCED, HANDOFF_FENCE_FREE = 0, 1
ARITH_EXACT, ARITH_CONTRACTED = 0, 1
SEED_KEEP_ACCUMULATOR, SEED_FROM_RESULT, SEED_AUTO = 0, 1, 2
FORM_AUTO, FORM_STREAM, FORM_PIXEL = 0, 1, 2
PTMI_OK, PTMI_EINVAL, PTMI_ENODEVICE, PTMI_EHIP, PTMI_ENOMEM, PTMI_ESTATE, PTMI_ELIMIT, PTMI_ESTALE = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/ptmi.h declares: name -> (restype, argtypes)
_f32p, _u32p, _i32p, _i64p, _vp = (C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int64), C.c_void_p)


class Stats(C.Structure):
    _fields_ = [("live_bounces", C.c_uint64), ("nominal_bounces", C.c_uint64), ("samples", C.c_uint64),
                ("last_render_ms", C.c_float), ("stream_iterations", C.c_uint32), ("stream_rays_dropped", C.c_uint64),
                ("stream_rays_truncated", C.c_uint64), ("stream_rays_spilled", C.c_uint64), ("stream_rays_overflowed", C.c_uint64)]


class ChainStats(C.Structure):
    _fields_ = [("states_on_device", C.c_uint32), ("states_on_host", C.c_uint32), ("device_slots", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("renders_chained", C.c_uint64), ("renders_in_place", C.c_uint64), ("renders_uploaded", C.c_uint64), ("evictions", C.c_uint64), ("fetches", C.c_uint64)]


SYMBOLS = {
    "ptmi_version": (C.c_int, []),
    "ptmi_build_id": (C.c_char_p, []),
    "ptmi_strerror": (C.c_char_p, [C.c_int]),
    "ptmi_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "ptmi_destroy": (None, [_vp]),
    "ptmi_last_error": (C.c_char_p, [_vp]),
    "ptmi_set_scene": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int]),
    "ptmi_resize": (C.c_int, [_vp, C.c_int, C.c_int]),
    "ptmi_set_partition": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "ptmi_local_rows": (C.c_int, [_vp]),
    "ptmi_global_row": (C.c_int, [_vp, C.c_int]),
    "ptmi_bind_planes": (C.c_int, [_vp] + [_vp] * 7),
    "ptmi_get_planes": (C.c_int, [_vp] + [C.POINTER(_vp)] * 7),
    "ptmi_set_stream": (C.c_int, [_vp, _vp]),
    "ptmi_set_timing": (C.c_int, [_vp, C.c_int]),
    "ptmi_set_variant": (C.c_int, [_vp, C.c_int]),
    "ptmi_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "ptmi_get_option": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)]),
    "ptmi_init_output": (C.c_int, [_vp, C.c_uint64]),
    "ptmi_reseed": (C.c_int, [_vp, C.c_uint64]),
    "ptmi_create_with": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ptmi_upload_state": (C.c_int, [_vp] + [_vp] * 7),
    "ptmi_download_state": (C.c_int, [_vp] + [_vp] * 7),
    "ptmi_download_color": (C.c_int, [_vp] + [_vp] * 3),
    "ptmi_render": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "ptmi_synchronize": (C.c_int, [_vp]),
    "ptmi_render_blocks": (C.c_int, [_vp, C.c_int]),
    "ptmi_render1": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp] + [_vp] * 14),
    "ptmi_render1_chained": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int] + [_vp] * 7 + [C.POINTER(C.c_uint64)] + [_vp] * 7),
    "ptmi_chain_init_output": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ptmi_chain_reseed": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "ptmi_chain_fetch": (C.c_int, [_vp, C.c_uint64] + [_vp] * 7),
    "ptmi_chain_release": (C.c_int, [_vp, C.c_uint64]),
    "ptmi_chain_info": (C.c_int, [_vp, C.POINTER(ChainStats)]),
    "ptmi_present": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "ptmi_snapshot_color": (C.c_int, [_vp, _vp, _vp]),
    "ptmi_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "ptmi_group_create": (C.c_int, [C.POINTER(_vp), _i32p, C.c_int, C.c_int]),
    "ptmi_group_destroy": (None, [_vp]),
    "ptmi_group_size": (C.c_int, [_vp]),
    "ptmi_group_member": (_vp, [_vp, C.c_int]),
    "ptmi_group_last_error": (C.c_char_p, [_vp]),
    "ptmi_group_set_scene": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int]),
    "ptmi_group_resize": (C.c_int, [_vp, C.c_int, C.c_int]),
    "ptmi_group_init_output": (C.c_int, [_vp, C.c_uint64]),
    "ptmi_group_reseed": (C.c_int, [_vp, C.c_uint64]),
    "ptmi_group_render": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "ptmi_group_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "ptmi_group_set_variant": (C.c_int, [_vp, C.c_int]),
    "ptmi_group_synchronize": (C.c_int, [_vp]),
    "ptmi_group_download_color": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ptmi_group_gather_color": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "ptmi_group_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "ptmi_partition_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ptmi_partition_global_row": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ptmi_reset_stats": (C.c_int, [_vp]),
    "ptmi_debug_counters": (C.c_int, [_vp, _vp]),
    "ptmi_debug_counters_n": (C.c_int, [_vp, _vp, C.c_int]),
    "ptmi_order_schedule": (C.c_int, [C.c_int, C.c_int, _i32p, _i32p]),
    "ptmi_stream_schedule": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, _vp, C.c_int]),
    "ptmi_stream_tickets": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "ptmi_eval_distance_to_sphere": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "ptmi_eval_distance_to_plane": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "ptmi_eval_sincos": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
}


class PtmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libptmi error %d: %s" % (code, message))
        self.code = code


_lib = None


def open_library(path, check_build_id=True):
    """dlopen one libptmi build and type every symbol (a second build -- the ablation library, a diagnostic build -- can be
    open beside the default one: Context(library=...)).
    The library must hold the code the sources beside this file compile to: either it was linked from this very text, or its
    ptmi_build_id() -- a hash over the compiled objects' code (_build.code_id) -- is what the sources give now (a comment edit
    keeps it).  A binary that travelled with edited kernels, or one from an older checkout, is refused here instead of being
    measured under the wrong name."""
    if not os.path.exists(path):
        raise PtmiError(PTMI_ESTATE, "libptmi.so not built (%s); run __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    lib.build_id = (lib.ptmi_build_id() or b"").decode("ascii", "replace")
    if check_build_id and not _build.matches_sources(path):
        raise PtmiError(PTMI_ESTATE, "%s was built from other sources: it carries build id %r (linked from text %r), the sources here "
                                     "have text hash %r and do not compile to that code; rebuild it (__graft_entry__.build())"
                        % (path, lib.build_id, _build.read_source_hash(path), _build.source_hash()))
    return lib


def load_library(path=None, check_build_id=True):
    """The default library of the process: libptmi.so (building nothing: see _build.build_lib), or `path` from then on.
    check_build_id=False only for tools that compare binaries of DIFFERENT sources on purpose (tools/ab.py prints each one's id)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    _lib = open_library(path or _build.LIB, check_build_id)
    return _lib


def stream_schedule(n_spp, n_pixels, lanes, batch=0, graded=True):
    """ptmi_stream_schedule: the first sample of every pass of the stream form's split kernel, and n_spp behind them (host arithmetic)."""
    first = np.zeros(65, np.int32)
    passes = load_library().ptmi_stream_schedule(int(n_spp), int(n_pixels), int(lanes), int(batch), 1 if graded else 0, _ptr(first), 65)
    if passes < 0:
        raise PtmiError(passes, "ptmi_stream_schedule")
    return [int(v) for v in first[:passes + 1]]


def stream_tickets(option, first, queue_regions):
    """ptmi_stream_tickets: [(pass, region index within the queue)] in the order one ticket queue of the split kernel hands them out, for
    the schedule `first` of stream_schedule."""
    passes = len(first) - 1
    n = passes * queue_regions
    f = np.asarray(first, np.int32)
    p, r = np.zeros(n, np.int32), np.zeros(n, np.int32)
    got = load_library().ptmi_stream_tickets(int(option), _ptr(f), passes, int(queue_regions), _ptr(p), _ptr(r), n)
    if got < 0:
        raise PtmiError(got, "ptmi_stream_tickets")
    return list(zip(p[:got].tolist(), r[:got].tolist()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _host(a, dtype, n=None, name="array"):
    a = np.ascontiguousarray(a, dtype=dtype)
    if n is not None and a.size != n:
        raise ValueError("%s has %d elements, expected %d" % (name, a.size, n))
    return a


class Context:
    """One ptmi_ctx.  Methods map 1:1 onto include/ptmi.h."""

    def __init__(self, device=0, library=None):
        self._lib = library if library is not None else load_library()
        h = _vp()
        rc = self._lib.ptmi_create(C.byref(h), int(device))
        if rc != PTMI_OK:
            raise PtmiError(rc, (self._lib.ptmi_last_error(None) or b"").decode())
        self._h = h
        self.width = self.height = 0
        self._keep = []   # keeps bound tensors alive

    # -- plumbing --------------------------------------------------------------------
    def _check(self, rc):
        if rc != PTMI_OK:
            raise PtmiError(rc, (self._lib.ptmi_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptmi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ---------------------------------------------------------------
    def set_scene(self, spheres, planes):
        s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        p = np.ascontiguousarray(planes, dtype=PLANE_DTYPE)
        self._check(self._lib.ptmi_set_scene(self._h, _ptr(s) if s.size else None, s.size,
                                             _ptr(p) if p.size else None, p.size))

    def set_partition(self, stripe_rows, n_parts, part):
        self._check(self._lib.ptmi_set_partition(self._h, stripe_rows, n_parts, part))

    def resize(self, width, height):
        self._check(self._lib.ptmi_resize(self._h, width, height))
        self.width, self.height = width, height
        self._keep = []

    @property
    def local_rows(self):
        rc = self._lib.ptmi_local_rows(self._h)
        if rc < 0:
            self._check(rc)
        return rc

    def global_row(self, local_row):
        rc = self._lib.ptmi_global_row(self._h, local_row)
        if rc < 0:
            raise PtmiError(rc, "ptmi_global_row")
        return rc

    def global_rows(self):
        return np.array([self.global_row(i) for i in range(self.local_rows)], dtype=np.int64)

    @property
    def n_local(self):
        return self.local_rows * self.width

    def bind_torch(self, color, state):
        """color: float32 CUDA tensor [3, rows, W]; state: int32 CUDA tensor [4, rows, W] (uint32 bits)."""
        assert color.is_cuda and state.is_cuda and color.is_contiguous() and state.is_contiguous()
        assert tuple(color.shape) == (3, self.local_rows, self.width), color.shape
        assert tuple(state.shape) == (4, self.local_rows, self.width), state.shape
        assert color.element_size() == 4 and state.element_size() == 4
        ptrs = [color[i].data_ptr() for i in range(3)] + [state[i].data_ptr() for i in range(4)]
        self._check(self._lib.ptmi_bind_planes(self._h, *[_vp(p) for p in ptrs]))
        self._keep = [color, state]

    def unbind(self):
        self._check(self._lib.ptmi_bind_planes(self._h, *([None] * 7)))
        self._keep = []

    def device_planes(self):
        """Device addresses (ints) of the seven planes the context renders into."""
        ptrs = [_vp() for _ in range(7)]
        self._check(self._lib.ptmi_get_planes(self._h, *[C.byref(p) for p in ptrs]))
        return [p.value for p in ptrs]

    def set_stream(self, hip_stream):
        self._check(self._lib.ptmi_set_stream(self._h, _vp(hip_stream) if hip_stream else None))

    def set_timing(self, enabled):
        self._check(self._lib.ptmi_set_timing(self._h, int(bool(enabled))))

    def set_variant(self, variant):
        self._check(self._lib.ptmi_set_variant(self._h, int(variant)))

    def set_option(self, option, value):
        self._check(self._lib.ptmi_set_option(self._h, int(option), C.c_int64(int(value))))

    def get_option(self, option):
        v = C.c_int64(0)
        self._check(self._lib.ptmi_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    # -- state -----------------------------------------------------------------------
    def init_output(self, seed0):
        self._check(self._lib.ptmi_init_output(self._h, C.c_uint64(seed0)))

    def reseed(self, seed0):
        self._check(self._lib.ptmi_reseed(self._h, C.c_uint64(seed0)))

    def create_with(self, w0, w1, w2):
        n = self.n_local
        ws = [_host(w, np.uint32, n, "word plane") for w in (w0, w1, w2)]
        self._check(self._lib.ptmi_create_with(self._h, *[_ptr(w) for w in ws]))

    def upload_state(self, r=None, g=None, b=None, sa=None, sb=None, sc=None, sctr=None):
        n = self.n_local
        arrs = [None if a is None else _host(a, np.float32, n) for a in (r, g, b)] + \
               [None if a is None else _host(a, np.uint32, n) for a in (sa, sb, sc, sctr)]
        self._check(self._lib.ptmi_upload_state(self._h, *[_ptr(a) for a in arrs]))

    def download_state(self):
        shape = (self.local_rows, self.width)
        out = [np.empty(shape, np.float32) for _ in range(3)] + [np.empty(shape, np.uint32) for _ in range(4)]
        self._check(self._lib.ptmi_download_state(self._h, *[_ptr(a) for a in out]))
        return tuple(out)

    def download_color(self):
        shape = (self.local_rows, self.width)
        out = [np.empty(shape, np.float32) for _ in range(3)]
        self._check(self._lib.ptmi_download_color(self._h, *[_ptr(a) for a in out]))
        return tuple(out)

    # -- hot path --------------------------------------------------------------------
    def render(self, camera, bounce_limit, n_spp, algorithm=INLINE):
        cam = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE)
        self._check(self._lib.ptmi_render(self._h, _ptr(cam), algorithm, bounce_limit, n_spp))

    def synchronize(self):
        self._check(self._lib.ptmi_synchronize(self._h))

    def render_blocks(self, algorithm=INLINE):
        rc = self._lib.ptmi_render_blocks(self._h, int(algorithm))
        if rc < 0:
            self._check(rc)
        return rc == 1

    def render1(self, camera, bounce_limit, width, height, planes_in, algorithm=INLINE, screen=None):
        """compileFor's closure (app/Main.hs:188-191): 7 host planes in -> 7 new host planes out."""
        cam = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE)
        n = width * height
        ins = [_host(a, np.float32, n) for a in planes_in[:3]] + [_host(a, np.uint32, n) for a in planes_in[3:]]
        outs = [np.empty((height, width), np.float32) for _ in range(3)] + \
               [np.empty((height, width), np.uint32) for _ in range(4)]
        sx = sy = None
        if screen is not None:
            sx, sy = _host(screen[0], np.int64, n), _host(screen[1], np.int64, n)
        self._check(self._lib.ptmi_render1(self._h, _ptr(cam), algorithm, bounce_limit, width, height,
                                           _ptr(sx), _ptr(sy), *[_ptr(a) for a in ins], *[_ptr(a) for a in outs]))
        return tuple(outs)

    # -- the closure, chained (device residency behind compileFor's pure type) ----------
    def render1_chained(self, camera, bounce_limit, width, height, token=0, planes_in=None, algorithm=INLINE, consume=False, fetch=()):
        """One application of the closure to the state `token` names (or, if the context does not hold it, to the seven host planes
        planes_in).  Returns (new token, {plane name: array} for the names in `fetch`, a subset of r g b sa sb sc sctr)."""
        cam = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE)
        n = width * height
        ins = [None] * 7
        if planes_in is not None:
            ins = [_host(a, np.float32, n) for a in planes_in[:3]] + [_host(a, np.uint32, n) for a in planes_in[3:]]
        names = "r g b sa sb sc sctr".split()
        outs = [np.empty((height, width), np.float32 if i < 3 else np.uint32) if names[i] in fetch else None for i in range(7)]
        tok = C.c_uint64(0)
        self._check(self._lib.ptmi_render1_chained(self._h, _ptr(cam), algorithm, bounce_limit, width, height, C.c_uint64(int(token)),
                                                   CHAIN_CONSUME if consume else 0, *[_ptr(a) for a in ins], C.byref(tok), *[_ptr(a) for a in outs]))
        return int(tok.value), {nm: a for nm, a in zip(names, outs) if a is not None}

    def chain_init_output(self, width, height, seed0):
        tok = C.c_uint64(0)
        self._check(self._lib.ptmi_chain_init_output(self._h, width, height, C.c_uint64(seed0), C.byref(tok)))
        return int(tok.value)

    def chain_reseed(self, seed0, width, height, token=0, colour_in=None, consume=False):
        ins = [None] * 3 if colour_in is None else [_host(a, np.float32, width * height) for a in colour_in]
        tok = C.c_uint64(0)
        self._check(self._lib.ptmi_chain_reseed(self._h, C.c_uint64(seed0), width, height, C.c_uint64(int(token)), CHAIN_CONSUME if consume else 0,
                                                *[_ptr(a) for a in ins], C.byref(tok)))
        return int(tok.value)

    def chain_fetch(self, token, width, height, planes="r g b sa sb sc sctr"):
        names = "r g b sa sb sc sctr".split()
        want = planes.split() if isinstance(planes, str) else list(planes)
        outs = [np.empty((height, width), np.float32 if i < 3 else np.uint32) if names[i] in want else None for i in range(7)]
        self._check(self._lib.ptmi_chain_fetch(self._h, C.c_uint64(int(token)), *[_ptr(a) for a in outs]))
        return tuple(a for a in outs if a is not None)

    def chain_release(self, token):
        self._check(self._lib.ptmi_chain_release(self._h, C.c_uint64(int(token))))

    def chain_info(self):
        st = ChainStats()
        self._check(self._lib.ptmi_chain_info(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in ChainStats._fields_}

    def present(self, iterations, rgb32f=True, rgba8=True):
        """graphicsLoop + fs.glsl: interleaved colour / iterations as float RGB and/or 8-bit RGBA."""
        shape = (self.local_rows, self.width)
        rgb = np.empty(shape + (3,), np.float32) if rgb32f else None
        rgba = np.empty(shape + (4,), np.uint8) if rgba8 else None
        self._check(self._lib.ptmi_present(self._h, int(iterations), _ptr(rgb), _ptr(rgba)))
        return rgb, rgba

    def stats(self):
        st = Stats()
        self._check(self._lib.ptmi_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in Stats._fields_}

    def debug_counters(self):
        out = np.zeros(256, np.uint32)
        n = self._lib.ptmi_debug_counters_n(self._h, _ptr(out), out.size)
        if n < 0:
            self._check(n)
        return out

    def reset_stats(self):
        self._check(self._lib.ptmi_reset_stats(self._h))

    # -- point queries ---------------------------------------------------------------
    def eval_distance_to_sphere(self, spheres, rays):
        s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        r = _host(rays, np.float32, s.size * 6, "rays")
        n = s.size
        just, t, nrm = np.empty(n, np.int32), np.empty(n, np.float32), np.empty((n, 6), np.float32)
        self._check(self._lib.ptmi_eval_distance_to_sphere(self._h, _ptr(s), _ptr(r), n, _ptr(just), _ptr(t), _ptr(nrm)))
        return just, t, nrm

    def eval_distance_to_plane(self, planes, rays):
        p = np.ascontiguousarray(planes, dtype=PLANE_DTYPE)
        r = _host(rays, np.float32, p.size * 6, "rays")
        n = p.size
        just, t, nrm = np.empty(n, np.int32), np.empty(n, np.float32), np.empty((n, 6), np.float32)
        self._check(self._lib.ptmi_eval_distance_to_plane(self._h, _ptr(p), _ptr(r), n, _ptr(just), _ptr(t), _ptr(nrm)))
        return just, t, nrm

    def eval_sincos(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        s, c = np.empty_like(x), np.empty_like(x)
        self._check(self._lib.ptmi_eval_sincos(self._h, _ptr(x), x.size, _ptr(s), _ptr(c)))
        return s, c


class Group:
    """One ptmi_group: the GPUs of a node behind one host process (include/ptmi.h, "groups")."""

    def __init__(self, devices, stripe_rows=0, library=None):
        self._lib = library if library is not None else load_library()
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = _vp()
        rc = self._lib.ptmi_group_create(C.byref(h), devs.ctypes.data_as(_i32p), devs.size, int(stripe_rows))
        if rc != PTMI_OK:
            raise PtmiError(rc, (self._lib.ptmi_last_error(None) or b"").decode())
        self._h = h
        self.width = self.height = 0

    def _check(self, rc):
        if rc != PTMI_OK:
            raise PtmiError(rc, (self._lib.ptmi_group_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptmi_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self._lib.ptmi_group_size(self._h)

    def member(self, i):
        """A non-owning Context view of member i (do not close it)."""
        c = Context.__new__(Context)
        c._lib, c._h, c._keep = self._lib, _vp(self._lib.ptmi_group_member(self._h, i)), []
        c.width, c.height = self.width, self.height
        c.close = lambda: None
        return c

    def set_scene(self, spheres, planes):
        s = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        p = np.ascontiguousarray(planes, dtype=PLANE_DTYPE)
        self._check(self._lib.ptmi_group_set_scene(self._h, _ptr(s) if s.size else None, s.size, _ptr(p) if p.size else None, p.size))

    def resize(self, width, height):
        self._check(self._lib.ptmi_group_resize(self._h, width, height))
        self.width, self.height = width, height

    def init_output(self, seed0):
        self._check(self._lib.ptmi_group_init_output(self._h, C.c_uint64(seed0)))

    def reseed(self, seed0):
        self._check(self._lib.ptmi_group_reseed(self._h, C.c_uint64(seed0)))

    def render(self, camera, bounce_limit, n_spp, algorithm=INLINE):
        cam = np.ascontiguousarray(camera, dtype=CAMERA_DTYPE)
        self._check(self._lib.ptmi_group_render(self._h, _ptr(cam), algorithm, bounce_limit, n_spp))

    def set_option(self, option, value):
        self._check(self._lib.ptmi_group_set_option(self._h, int(option), C.c_int64(int(value))))

    def set_variant(self, variant):
        self._check(self._lib.ptmi_group_set_variant(self._h, int(variant)))

    def synchronize(self):
        self._check(self._lib.ptmi_group_synchronize(self._h))

    def download_color(self):
        out = [np.empty((self.height, self.width), np.float32) for _ in range(3)]
        self._check(self._lib.ptmi_group_download_color(self._h, *[_ptr(a) for a in out]))
        return tuple(out)

    def gather_color(self, root, r_dev, g_dev, b_dev):
        """r_dev, g_dev, b_dev: device pointers (int) of [height][width] float planes on member `root`'s device."""
        self._check(self._lib.ptmi_group_gather_color(self._h, int(root), _vp(r_dev), _vp(g_dev), _vp(b_dev)))

    def stats(self):
        st = Stats()
        self._check(self._lib.ptmi_group_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in Stats._fields_}
