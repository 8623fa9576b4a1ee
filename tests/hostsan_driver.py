"""Driver of tests/test_host_sanitized.py (run in a CHILD process with the HIP stand-in and the sanitizer runtime preloaded; not a test
module itself).  It takes the sanitized libptmi through scenarios of C-ABI calls -- the resident path, GLASS and the stream form, the
closures, a partitioned image, host threads, a group with the RCCL stand-in -- first plainly, then once per failure point: the k-th hipMalloc /
copy / launch / synchronize / pinned allocation / stream-or-event creation of the scenario fails, for every k (alone; with the next call of
the kind; with every later one).  After every run every context is destroyed and the stand-in must hold no block, pinned block, stream or
event.  Kernels do not run on the stand-in: no value is checked."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
B = pkg.binding
stub = ctypes.CDLL(os.environ["PTMI_HIPSTUB"])
assert stub.hipstub_is_the_stub() == 1
stub.hipstub_live_bytes.restype = ctypes.c_ulonglong
stub.hipstub_fail.argtypes = [ctypes.c_int, ctypes.c_long]
stub.hipstub_fail_run.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_long]
stub.hipstub_calls.restype = ctypes.c_long
stub.hipstub_set_device_size.argtypes = [ctypes.c_int, ctypes.c_ulonglong]
stub.hipstub_poke.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_ulonglong, ctypes.c_uint]
stub.hipstub_launches.argtypes = [ctypes.c_char_p]
stub.hipstub_launches.restype = ctypes.c_long
B.load_library(os.environ["PTMI_SANITIZED_LIB"], check_build_id=os.environ.get("PTMI_HOSTSAN_FOREIGN_BUILD") != "1")     # (1: a gcov build made by hand)
assert b"fsanitize" in B.load_library().ptmi_build_id() or True      # (the id carries a suffix for the extra flags; not relied upon)

KINDS = ["hipMalloc", "copy", "launch", "synchronize", "hipHostMalloc", "stream/event creation"]
MORE = [0, 1, 1 << 30]
MORE_STRIDE = int(os.environ.get("PTMI_HOSTSAN_MORE_STRIDE", "3"))
RATE = float(os.environ.get("PTMI_HOSTSAN_RATE", "0.03"))          # the random walk: how often a step arms an injected failure
PLAIN = True                                                        # False while a failure is being injected: a scenario's own expectations hold for the plain run only
cam = pkg.world.initial_camera()
cam2 = cam.copy()
cam2["position"][0] += 0.25


def planes(w, h, seed=3):
    r = np.random.default_rng(seed)
    return [np.zeros((h, w), np.float32) for _ in range(3)] + [r.integers(0, 2 ** 32, (h, w), dtype=np.uint32) for _ in range(4)]


def dev_alloc(n_bytes):
    p = ctypes.c_void_p()
    rc = stub.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n_bytes))
    if rc != 0:
        stub.hipstub_clear_error()                                  # (the caller's own failed call: the caller's to clear)
        raise MemoryError("stand-in hipMalloc (injected)")
    return p.value


def dev_free(p):
    stub.hipFree(ctypes.c_void_p(p))


def quiet(call):
    """Clean-up calls after a (possibly injected) failure: their own failure is not the scenario's."""
    try:
        call()
    except B.PtmiError:
        pass


class DeviceBlocks:
    """The caller's own device buffers (an injected failure may hit one of these allocations too: the earlier ones are freed)."""

    def __init__(self, sizes):
        self.sizes, self.ptrs = list(sizes), []

    def __enter__(self):
        try:
            for n in self.sizes:
                self.ptrs.append(dev_alloc(n))
        except MemoryError:
            self.__exit__()
            raise
        return self.ptrs

    def __exit__(self, *exc):
        for p in self.ptrs:
            dev_free(p)
        self.ptrs = []


# ---- scenarios: each makes and destroys its own contexts ------------------------------------------------------------------------------
def resident():
    sp, pl = pkg.world.scene16()
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.set_timing(True)
        for (w, h) in ((72, 40), (136, 24), (8, 8)):               # grow, reshape, shrink
            ctx.resize(w, h)
            ctx.init_output(7)
            for alg in (pkg.INLINE, pkg.STREAMS):
                for spp in (1, 4, 4, 4, 4, 4):                       # (the same key five times: the cost order is rebuilt before launch 1, 2, 4)
                    ctx.render(cam, 8, spp, alg)
            ctx.render(cam2, 0, 2)
            ctx.render(cam2, 15, 0)
            ctx.reseed(9)
            ctx.synchronize()
            ctx.download_color()
            st = ctx.download_state()
            ctx.upload_state(*st)
            ctx.upload_state(r=st[0])
            ctx.create_with(st[3], st[4], st[5])
            ctx.present(3)
            ctx.present(1, rgb32f=False)
            ctx.stats(); ctx.debug_counters(); ctx.reset_stats()
            ctx.render_blocks(pkg.INLINE); ctx.render_blocks(pkg.STREAMS)
            ctx.device_planes()
        for opt, val in ((B.OPT_SPP_CHUNKS, 4), (B.OPT_STREAMS_SEED_RULE, B.SEED_FROM_RESULT), (B.OPT_STREAM_STEP_CAP, 5), (B.OPT_ARITHMETIC, B.ARITH_CONTRACTED)):
            ctx.set_option(opt, val)
            assert ctx.get_option(opt) == val
            ctx.render(cam, 4, 2, pkg.INLINE)
            ctx.render(cam, 4, 2, pkg.STREAMS)
        s3 = sp[:3]
        ctx.eval_distance_to_sphere(s3, np.ones((3, 6), np.float32))
        ctx.eval_distance_to_plane(pl, np.ones((pl.size, 6), np.float32))
        ctx.eval_sincos(np.linspace(-4, 4, 100).astype(np.float32))
        # a caller's stream and bound planes
        s = ctypes.c_void_p()
        if stub.hipStreamCreateWithFlags(ctypes.byref(s), 1) != 0:
            stub.hipstub_clear_error()
            raise MemoryError("stand-in hipStreamCreateWithFlags (injected)")
        try:
            ctx.set_stream(s.value)
            ctx.resize(40, 16)
            ctx.init_output(1)
            ctx.render(cam, 8, 2)
            ctx.synchronize()
            ctx.set_stream(None)
        finally:
            quiet(ctx.synchronize)
            for _ in range(2):
                try:
                    ctx.set_stream(None)
                    break
                except B.PtmiError:
                    pass
            else:
                ctx.close()                                         # the context could not be told (the device stays broken): it goes before the stream does
            stub.hipStreamDestroy(s)
        with DeviceBlocks([40 * 16 * 4] * 7) as bound:
            try:
                ctx._check(ctx._lib.ptmi_bind_planes(ctx._h, *[ctypes.c_void_p(p) for p in bound]))
                ctx.init_output(2)
                ctx.render(cam, 8, 2)
                ctx.download_color()
                ctx.unbind()
                ctx.render(cam, 8, 1)
            finally:
                quiet(ctx.synchronize)
                quiet(ctx.unbind)


def partitioned():
    sp, pl = pkg.world.main_scene()
    for n_parts, part, stripe in ((3, 0, 8), (3, 2, 8), (8, 7, 2), (2, 1, 16)):
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.set_partition(stripe, n_parts, part)
            ctx.resize(64, 50)
            assert ctx.local_rows == B.load_library().ptmi_partition_rows(50, stripe, n_parts, part)
            ctx.global_rows()
            ctx.init_output(5)
            ctx.render(cam, 8, 3)
            ctx.render(cam, 8, 300, pkg.STREAMS)                     # (no GLASS: the per-pixel chain whatever the sample count)
            ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            ctx.render(cam, 8, 300, pkg.STREAMS)
            ctx.render(cam, 8, 2, pkg.STREAMS)
            ctx.download_color()
            ctx.present(2)
            with DeviceBlocks([3 * max(ctx.local_rows, 1) * 64 * 4]) as (dst,):
                try:
                    ctx._check(ctx._lib.ptmi_snapshot_color(ctx._h, ctypes.c_void_p(dst), None))
                finally:
                    quiet(ctx.synchronize)


def glass():
    sp, pl = pkg.world.glass_scene()
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.resize(80, 48)
        ctx.init_output(11)
        ctx.render(cam, 8, 4, pkg.STREAMS)                          # tree walk
        for form in (B.FORM_STREAM, B.FORM_PIXEL, B.FORM_AUTO):
            ctx.set_option(B.OPT_STREAMS_FORM, form)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            ctx.render(cam, 8, 4, pkg.STREAMS)
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        for opt, val in ((B.OPT_STREAM_CAPACITY, 1), (B.OPT_STREAM_BATCH, 2), (B.OPT_GLASS_BATCH, 8), (B.OPT_STREAM_GRADED, 0), (B.OPT_STREAM_PASS_GROUPS, 2),
                         (B.OPT_STREAM_TAIL, 200), (B.OPT_SNAPSHOT_BUDGET_MB, 1), (B.OPT_ORDERED_PASSES, 1), (B.OPT_PASS_HANDOFF, B.HANDOFF_FENCE_FREE),
                         (B.OPT_ORDERED_PASSES, 2), (B.OPT_STREAM_STEP_CAP, 3)):
            ctx.set_option(opt, val)
            ctx.render(cam, 8, 70, pkg.STREAMS)
        ctx.stats()
        ctx.resize(24, 200)
        ctx.init_output(12)
        ctx.render(cam2, 8, 260, pkg.STREAMS)
        try:
            ctx.render(cam, 8, 1, pkg.INLINE)                       # refused: Inline cannot split rays
            raise AssertionError("render Inline with GLASS was accepted")
        except B.PtmiError as e:
            assert e.code == B.PTMI_EINVAL if hasattr(B, "PTMI_EINVAL") else True
    with pkg.Context(0) as ctx:                                     # one part of a partitioned glass image at >= 256 spp: FORM_AUTO = the stream form, ordered passes
        ctx.set_scene(sp, pl)
        ctx.set_partition(8, 4, 1)
        ctx.resize(96, 64)
        ctx.init_output(13)
        ctx.render(cam, 8, 256, pkg.STREAMS)
        ctx.render(cam, 8, 256, pkg.STREAMS)
        ctx.render(cam, 8, 8, pkg.STREAMS)
        ctx.set_scene(*pkg.world.scene16())                         # the scene loses its GLASS
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        ctx.render(cam, 8, 8, pkg.STREAMS)
        ctx.render(cam, 8, 8, pkg.INLINE)


def closures():
    sp, pl = pkg.world.scene16()
    w, h = 48, 20
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        p = planes(w, h)
        out = ctx.render1(cam, 8, w, h, p)
        out = ctx.render1(cam, 8, w, h, out, pkg.STREAMS)
        ys, xs = np.mgrid[0:h, 0:w]
        ctx.render1(cam, 8, w, h, out, screen=(xs.astype(np.int64), ys.astype(np.int64)))
        ctx.render1(cam, 8, 2 * w, h, planes(2 * w, h))            # another size
        ctx.set_option(B.OPT_CHAIN_SLOTS, 3)
        t0 = ctx.chain_init_output(w, h, 21)
        toks = [t0]
        for i in range(12):                                         # tokens pile up: evictions to host memory
            t, got = ctx.render1_chained(cam if i % 5 else cam2, 8, w, h, token=toks[-1], fetch=("r", "g", "b") if i % 4 == 0 else ())
            toks.append(t)
        info = ctx.chain_info()
        assert info["states_on_device"] <= 3 and info["states_on_host"] >= 1, info
        ctx.chain_fetch(toks[1], w, h)                              # from the host copy
        ctx.chain_fetch(toks[-1], w, h, "sa sctr")
        t, _ = ctx.render1_chained(cam, 8, w, h, token=toks[2])    # an evicted state is the input: back to the device
        t, _ = ctx.render1_chained(cam, 8, w, h, token=toks[3], consume=True)
        t, _ = ctx.render1_chained(cam, 8, w, h, token=t, consume=True, algorithm=pkg.STREAMS, fetch=("sctr",))
        t = ctx.chain_reseed(5, w, h, token=t)
        t = ctx.chain_reseed(6, w, h, token=t, consume=True)
        t = ctx.chain_reseed(7, w, h, token=toks[4])                # evicted input
        t = ctx.chain_reseed(8, w, h, colour_in=p[:3])             # from host planes
        for k in toks[::2]:
            ctx.chain_release(k)
        ctx.chain_release(toks[0])                                  # twice: fine
        for call in (lambda: ctx.chain_fetch(toks[0], w, h), lambda: ctx.render1_chained(cam, 8, w, h, token=toks[0]),
                     lambda: ctx.render1_chained(cam, 8, w, h), lambda: ctx.render1_chained(cam, 8, w + 1, h, token=t),
                     lambda: ctx.chain_reseed(1, w, h, token=12345)):
            try:
                call()
                raise AssertionError("a stale or missing input was accepted")
            except B.PtmiError:
                pass
        t2, _ = ctx.render1_chained(cam, 8, w, h, token=toks[0], planes_in=p)      # stale token + host planes: the copy path
        w2, h2 = 30, 30                                             # another image size: the free blocks of the old size go
        u = ctx.chain_init_output(w2, h2, 1)
        for i in range(5):
            u, _ = ctx.render1_chained(cam, 8, w2, h2, token=u, consume=bool(i % 2))
        ctx.chain_fetch(u, w2, h2, "r")
        ctx.set_option(B.OPT_CHAIN_SLOTS, 0)
        ctx.chain_info()
        # the resident state is untouched by all of this
        ctx.resize(w, h)
        ctx.init_output(1)
        ctx.render(cam, 8, 2)
        ctx.download_color()
    with pkg.Context(0) as ctx:                                     # GLASS through the closures
        ctx.set_scene(*pkg.world.glass_scene())
        t = ctx.chain_init_output(w, h, 2)
        for _ in range(3):
            t, _ = ctx.render1_chained(cam, 8, w, h, token=t, algorithm=pkg.STREAMS)
        ctx.render1(cam, 8, w, h, planes(w, h), pkg.STREAMS)


def staged():
    """Copies of a megabyte and more go through the pinned ring and its worker threads (csrc/ptmi_stage.cpp)."""
    sp, pl = pkg.world.scene16()
    w, h = 320, 240                                                 # 7 planes = 2.1 MB
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        out = ctx.render1(cam, 8, w, h, planes(w, h))
        t, got = ctx.render1_chained(cam, 8, w, h, planes_in=out, fetch="r g b sa sb sc sctr".split())
        ctx.chain_fetch(t, w, h)
        ctx.set_option(B.OPT_CHAIN_SLOTS, 2)
        t2, _ = ctx.render1_chained(cam, 8, w, h, token=t)
        t3, _ = ctx.render1_chained(cam, 8, w, h, token=t2)        # t leaves the device through the ring
        ctx.chain_fetch(t, w, h, "r g b")
        ctx.resize(w, h)
        ctx.init_output(3)
        ctx.render(cam, 8, 1)
        st = ctx.download_state()
        ctx.upload_state(*st)
        ctx.download_color()
        ctx.present(1)
    with pkg.Group([0, 0], 8) as g:
        g.set_scene(sp, pl)
        g.resize(w, 2 * h)
        g.init_output(1)
        g.render(cam, 8, 1)
        g.download_color()


def threads():
    """Several host threads on one context (the application's computation, graphics and input threads: app/Main.hs:178-180), the group's
    per-member threads, the pinned ring's workers -- what ThreadSanitizer looks at (tests/test_host_sanitized.py, the second build)."""
    import threading
    sp, pl = pkg.world.scene16()
    w, h = 320, 240
    errors = []
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.resize(w, h)
        ctx.init_output(1)
        ctx.set_option(B.OPT_CHAIN_SLOTS, 3)
        start = ctx.chain_init_output(w, h, 5)
        held, lock = [start], threading.Lock()

        def closure_calls():
            try:
                t = start
                for i in range(6):
                    nxt, _ = ctx.render1_chained(cam, 8, w, h, token=t, fetch=("r",) if i % 2 else ())
                    if t != start:
                        with lock:
                            held.append(t)                            # (only states this thread has moved on from are the releaser's to take)
                    t = nxt
            except Exception as e:                                   # noqa: BLE001
                errors.append(e)

        def readers():
            try:
                for _ in range(6):
                    with lock:
                        t = held[len(held) // 2]
                    try:
                        ctx.chain_fetch(t, w, h, "r g b")
                    except B.PtmiError as e:
                        if e.code != B.PTMI_ESTALE:
                            raise
                    ctx.render(cam, 8, 1)
                    ctx.download_color()
                    ctx.stats()
            except Exception as e:                                   # noqa: BLE001
                errors.append(e)

        def releaser():
            try:
                for _ in range(8):
                    with lock:
                        t = held.pop(1) if len(held) > 2 else 0
                    if t:
                        ctx.chain_release(t)
                    ctx.chain_info()
            except Exception as e:                                   # noqa: BLE001
                errors.append(e)

        ts = [threading.Thread(target=f) for f in (closure_calls, closure_calls, readers, releaser)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    with pkg.Group([0, 0, 0], 8) as g:
        g.set_scene(sp, pl)
        g.resize(w, 2 * h)
        g.init_output(1)
        g.render(cam, 8, 1)
        g.download_color()                                           # one host thread per member
        with DeviceBlocks([w * 2 * h * 4] * 3) as dst:
            g.gather_color(1, *dst)
            quiet(g.synchronize)
    if errors:
        raise errors[0]


def same(got, want, what):
    for a, b in zip(got, want):
        assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)), what


def roundtrips():
    """What CAN be checked for value on the stand-in: everything that is copies only.  Kernels do not run, so a state that was uploaded, copied on
    the "device", moved to host memory and back, or snapshotted must come down as it went up -- in deferred mode (main: PTMI_HOSTSAN_DEFERRED)
    only if the library really synchronises before it reads a result, reuses a pinned buffer or returns a borrowed one."""
    sp, pl = pkg.world.scene16()
    for (w, h) in ((48, 20), (320, 240)):                           # (the second: through the pinned ring and its worker threads)
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.resize(w, h)
            p = planes(w, h, 5)
            p[0][:] = np.arange(w * h, dtype=np.float32).reshape(h, w)
            p[1][:] = 2.5; p[2][:] = -1.0
            ctx.upload_state(*p)
            same(ctx.download_state(), p, "upload_state / download_state")
            same(ctx.download_color(), p[:3], "download_color")
            q = planes(w, h, 6)
            ctx.upload_state(sa=q[3], sctr=q[6])                    # two planes only
            same(ctx.download_state(), p[:3] + [q[3], p[4], p[5], q[6]], "partial upload")
            with DeviceBlocks([3 * w * h * 4]) as (dst,):
                ctx._check(ctx._lib.ptmi_snapshot_color(ctx._h, ctypes.c_void_p(dst), None))
                ctx.synchronize()
                back = np.empty((3, h, w), np.float32)
                if stub.hipMemcpy(back.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dst), ctypes.c_size_t(back.nbytes), 2) != 0:
                    stub.hipstub_clear_error()
                    raise MemoryError("stand-in hipMemcpy (injected)")
                same(list(back), p[:3], "ptmi_snapshot_color")
            # the chained closure with kernels that do nothing: every state equals the planes the chain began with
            ctx.set_option(B.OPT_CHAIN_SLOTS, 2)
            t1, got = ctx.render1_chained(cam, 8, w, h, planes_in=p, fetch="r g b sa sb sc sctr".split())
            same([got[k] for k in "r g b sa sb sc sctr".split()], p, "chained: planes out of the uploading call")
            t2, _ = ctx.render1_chained(cam, 8, w, h, token=t1)
            t3, _ = ctx.render1_chained(cam, 8, w, h, token=t2)     # t1 moves to host memory
            assert ctx.chain_info()["states_on_host"] >= 1
            for t in (t1, t2, t3):
                same(ctx.chain_fetch(t, w, h), p, "chain_fetch")
            t4, _ = ctx.render1_chained(cam, 8, w, h, token=t1, consume=True)     # back from host memory, in place
            same(ctx.chain_fetch(t4, w, h), p, "consumed evicted state")
            t5 = ctx.chain_reseed(3, w, h, colour_in=p[:3])
            same(ctx.chain_fetch(t5, w, h, "r g b"), p[:3], "chain_reseed from host colour")
            same(ctx.render1(cam, 8, w, h, p), p, "copying closure")
    with pkg.Group([0, 0, 0], 4) as g:                              # the group's host read-out stitches what the members hold
        w, h = 40, 30
        g.set_scene(sp, pl)
        g.resize(w, h)
        whole = [np.random.default_rng(k).random((h, w), dtype=np.float32) for k in range(3)]
        for i in range(g.size):
            m = g.member(i)
            rows = m.global_rows()
            z = [np.zeros((rows.size, w), np.uint32)] * 4
            m.width, m.height = w, h
            m.upload_state(*[a[rows] for a in whole], *z)
        same(g.download_color(), whole, "group download_color")


def layout():
    """Where the stream form keeps its counters (csrc/ptmi_kernels.h): line L of the block is word L * kCounterStride."""
    import re
    text = open(os.path.join(ROOT, "haskell-path-tracer_amd", "csrc", "ptmi_kernels.h")).read()
    k = {}
    for stmt in re.findall(r"constexpr int ([^;]+);", text):
        for part in re.split(r",\s*(?=k[A-Z])", stmt):
            m = re.match(r"\s*(k\w+)\s*=\s*([^,;/]+)", part)
            if m:
                try:
                    k[m.group(1)] = int(eval(m.group(2), {}, dict(k)))
                except Exception:                                     # noqa: BLE001  (constants this function has no use for)
                    pass
    return k


def stream_overflow():
    """The stream form's host loop acts on what its kernels counted -- children left in the overflow stream (another LEVEL is launched), children
    that found it full (the call is REDONE with longer streams, the planes put back).  Kernels do not run here, so the counts are placed
    into the read-backs (hipstub_poke)."""
    k = layout()
    line = lambda n: n * k["kCounterStride"] * 4                   # noqa: E731  byte offset of counter line n
    level_line = lambda lv, i: line(k["kLvCursor"] + k["kLvPerLevel"] * lv + i)      # noqa: E731
    split, level = b"streams_split_kernel", b"streams_level_kernel"
    sp, pl = pkg.world.glass_scene()
    stub.hipstub_clear_pokes()
    try:
        with pkg.Context(0) as ctx:
            ctx.set_scene(sp, pl)
            ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            ctx.resize(160, 96)
            ctx.init_output(3)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            count = lambda name: stub.hipstub_launches(name)       # noqa: E731
            # 1. the split kernel left 3 children in its overflow stream: one level is launched for them
            n_split, n_level = count(split), count(level)
            ctx.reset_stats()
            stub.hipstub_poke(split, 1, level_line(0, 0), 3)        # the stream's cursor
            stub.hipstub_poke(split, 1, level_line(0, 2), 3)        # the children stored (shard 0)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            if PLAIN:
                assert (count(split) - n_split, count(level) - n_level) == (1, 1)
                assert ctx.stats()["stream_rays_overflowed"] == 3, ctx.stats()
            # 2. ... and that level leaves 2 of its own: a second level
            n_split, n_level = count(split), count(level)
            stub.hipstub_poke(split, 1, level_line(0, 0), 3)
            stub.hipstub_poke(split, 1, level_line(0, 3), 3)        # (another shard)
            stub.hipstub_poke(level, 1, level_line(1, 0), 2)
            stub.hipstub_poke(level, 1, level_line(1, 2), 2)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            if PLAIN:
                assert (count(split) - n_split, count(level) - n_level) == (1, 2)
            # 3. five children found the stream full: longer streams, the planes put back, the launch once more -- nothing is dropped
            n_split = count(split)
            ctx.reset_stats()
            stub.hipstub_poke(split, 1, line(k["kLvDropped"]), 5)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            if PLAIN:
                assert count(split) - n_split == 2
                assert ctx.stats()["stream_rays_dropped"] == 0, ctx.stats()
            # ... twice in one call
            n_split = count(split)
            stub.hipstub_poke(split, 1, line(k["kLvDropped"]), 5)
            stub.hipstub_poke(split, 2, line(k["kLvDropped"]), 1)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            if PLAIN:
                assert count(split) - n_split == 3
            # 4. with the streams at their longest the drops stand, counted
            ctx.set_option(B.OPT_STREAM_CAPACITY, 64)
            n_split = count(split)
            ctx.reset_stats()
            stub.hipstub_poke(split, 1, line(k["kLvDropped"]), 9)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            if PLAIN:
                assert count(split) - n_split == 1
                assert ctx.stats()["stream_rays_dropped"] == 9, ctx.stats()
            # 5. a smaller image afterwards, and the scene losing its GLASS (the colour backup goes)
            ctx.resize(64, 32)
            ctx.init_output(4)
            stub.hipstub_poke(split, 1, line(k["kLvDropped"]), 2)
            ctx.render(cam, 8, 4, pkg.STREAMS)
            ctx.set_scene(*pkg.world.scene16())
            ctx.render(cam, 8, 4, pkg.STREAMS)
    finally:
        stub.hipstub_clear_pokes()


def monkey(seed=0, steps=1500):
    """A seeded random walk over the C ABI on one context and one group: any order of calls with any arguments, valid or not, and now and then
    an injected runtime failure.  A refusal (PtmiError) is an answer; a sanitizer report, a hang or a leak is not."""
    r = np.random.default_rng(seed)
    scenes = [pkg.world.scene16(), pkg.world.main_scene(), pkg.world.glass_scene()]
    sizes = [(8, 8), (64, 16), (72, 40), (136, 24), (33, 7), (1, 1), (256, 64), (320, 240)]       # (the last: copies go through the pinned ring)
    absurd = [(2 ** 31 - 1, 2 ** 31 - 1), (1 << 20, 1 << 21), (2 ** 31 - 1, 1 << 12)]               # refused by count, or more than the device has: never wrapped
    k = layout()
    state = {"w": 0, "h": 0, "tokens": [], "glass": False}
    cams = [cam, cam2]

    def pick(xs):
        return xs[int(r.integers(0, len(xs)))]

    def wh():
        return (state["w"], state["h"]) if state["w"] and r.random() < 0.8 else pick(sizes)

    def wild_token():
        return int(r.integers(1, 1 << 50)) if r.random() < 0.5 else 0

    stream = ctypes.c_void_p()
    if stub.hipStreamCreateWithFlags(ctypes.byref(stream), 1) != 0:
        stub.hipstub_clear_error()
        return
    with DeviceBlocks([320 * 240 * 4] * 7) as mine, pkg.Context(0) as ctx, pkg.Group([0] * int(r.integers(1, 5)), int(pick([0, 2, 8]))) as g:
        def bind():
            if r.random() < 0.5:
                ctx._check(ctx._lib.ptmi_bind_planes(ctx._h, *[ctypes.c_void_p(p) for p in mine]))
            else:
                ctx.unbind()

        def streams():
            ctx.set_stream(stream.value if r.random() < 0.5 else None)

        def poke():                                                 # what the stream form's kernels "counted", for the next read-back
            which = pick([k["kLvDropped"], k["kLvCursor"], k["kLvCursor"] + 2, k["kLvCursor"] + k["kLvPerLevel"] + 2, k["kLvSplitPixels"], k["kLvDeepest"]])
            stub.hipstub_poke(pick([b"streams_split_kernel", b"streams_level_kernel", b"streams_primary_kernel"]), 1,
                              which * k["kCounterStride"] * 4, int(pick([1, 3, 64, 5000, 1 << 31, 0xffffffff])))

        def stream_form():
            ctx.set_option(B.OPT_STREAMS_FORM, int(pick([B.FORM_STREAM, B.FORM_STREAM, B.FORM_AUTO, B.FORM_PIXEL])))

        def set_scene():
            k = int(r.integers(0, len(scenes)))
            ctx.set_scene(*scenes[k]); state["glass"] = k == 2

        def resize():
            if r.random() < 0.1:
                try:
                    ctx.resize(*pick(absurd))
                    raise RuntimeError("an absurd size was accepted")
                except (B.PtmiError, MemoryError):
                    state["w"] = state["h"] = 0                     # (the old planes went first: the context is unsized now)
                    return
            w, h = pick(sizes)
            ctx.resize(w, h); state["w"], state["h"] = w, h

        def partition():
            n = int(pick([1, 2, 3, 8])); ctx.set_partition(int(pick([1, 2, 8, 16])), n, int(r.integers(0, n)))

        def option():
            ctx.set_option(int(r.integers(0, 17)), int(pick([-1, 0, 1, 2, 3, 4, 8, 64, 300, 1 << 20])))

        def render():
            ctx.render(pick(cams), int(pick([0, 1, 4, 8, 15])), int(pick([0, 1, 2, 5, 64, 260])), int(pick([pkg.INLINE, pkg.STREAMS])))

        def state_io():
            st = ctx.download_state(); ctx.upload_state(*st); ctx.download_color(); ctx.present(int(pick([0, 1, 7])))

        def seeds():
            pick([lambda: ctx.init_output(int(r.integers(0, 1 << 62))), lambda: ctx.reseed(int(r.integers(0, 1 << 62)))])()

        def closure_copying():
            w, h = wh()
            ctx.render1(pick(cams), 8, w, h, planes(w, h, int(r.integers(0, 9))), int(pick([pkg.INLINE, pkg.STREAMS])))

        def chain_new():
            w, h = wh()
            state["tokens"].append((ctx.chain_init_output(w, h, int(r.integers(0, 1 << 40))), w, h))

        def chain_call():
            tok, w, h = pick(state["tokens"]) if state["tokens"] and r.random() < 0.9 else (wild_token(),) + wh()
            if r.random() < 0.1:
                w += 1
            consume = r.random() < 0.3
            new, _ = ctx.render1_chained(pick(cams), 8, w, h, token=tok, consume=consume, algorithm=int(pick([pkg.INLINE, pkg.STREAMS])),
                                         planes_in=planes(w, h) if r.random() < 0.15 else None, fetch=pick([(), ("r",), ("r", "g", "b"), ("sa", "sctr")]))
            if consume:
                state["tokens"] = [x for x in state["tokens"] if x[0] != tok]
            state["tokens"].append((new, w, h))

        def chain_misc():
            if not state["tokens"]:
                return
            tok, w, h = pick(state["tokens"])
            k = int(r.integers(0, 4))
            if k == 0:
                ctx.chain_fetch(tok, w, h, pick(["r g b", "sa sb sc sctr", "r g b sa sb sc sctr", "b"]))
            elif k == 1:
                ctx.chain_release(tok); state["tokens"] = [x for x in state["tokens"] if x[0] != tok]
            elif k == 2:
                state["tokens"].append((ctx.chain_reseed(int(r.integers(0, 1 << 40)), w, h, token=tok), w, h))
            else:
                ctx.chain_info()

        def group_ops():
            k = int(r.integers(0, 6))
            if k == 0:
                g.set_scene(*pick(scenes[:2]))
            elif k == 1:
                g.resize(*pick(sizes))
            elif k == 2:
                g.init_output(3)
            elif k == 3:
                g.render(pick(cams), 8, int(pick([1, 2, 5])), int(pick([pkg.INLINE, pkg.STREAMS])))
            elif k == 4:
                g.download_color(); g.stats()
            elif g.width:
                with DeviceBlocks([g.width * g.height * 4] * 3) as dst:
                    try:
                        g.gather_color(int(r.integers(0, g.size)), *dst)
                    finally:
                        quiet(g.synchronize)

        def misc():
            pick([ctx.stats, ctx.debug_counters, ctx.reset_stats, ctx.synchronize, lambda: ctx.set_timing(bool(r.integers(0, 2))),
                  lambda: ctx.render_blocks(pkg.INLINE), lambda: ctx.set_variant(int(pick([0, 0, 4, 5, 13, 17, 99]))), ctx.device_planes])()

        menu = [set_scene, resize, partition, option, render, render, render, state_io, seeds, closure_copying, chain_new, chain_call, chain_call,
                chain_call, chain_misc, chain_misc, group_ops, group_ops, misc, bind, streams, poke, stream_form]
        for _ in range(steps):
            if r.random() < RATE:
                stub.hipstub_fail_run(int(r.integers(0, 6)), int(r.integers(1, 6)), int(pick([0, 0, 1, 3, 40])))
            try:
                pick(menu)()
            except (B.PtmiError, MemoryError):
                pass
            except AssertionError:
                pass                                                # (the binding's own argument checks)
            if len(state["tokens"]) > 40:
                for tok, _, _ in state["tokens"][:20]:
                    quiet(lambda: ctx.chain_release(tok))
                state["tokens"] = state["tokens"][20:]
        for kind in range(6):
            stub.hipstub_fail(kind, 0)
        stub.hipstub_clear_pokes()
        quiet(ctx.synchronize)
        quiet(g.synchronize)
        quiet(lambda: ctx.set_stream(None))
        quiet(ctx.unbind)
    stub.hipStreamDestroy(stream)
    stub.hipstub_clear_error()


def monkey_threads(seed=0, steps=300, n_threads=3):
    """The random walk from several threads at once on ONE context of a fixed size (the application's threads share a handle): renders, both
    closures, fetches and releases of each other's tokens, state transfers, statistics, failing calls and their messages."""
    import threading
    w, h = 72, 40
    sp, pl = pkg.world.scene16()
    problems = []
    with pkg.Context(0) as ctx:
        ctx.set_scene(sp, pl)
        ctx.resize(w, h)
        ctx.init_output(seed)
        ctx.set_option(B.OPT_CHAIN_SLOTS, 4)
        tokens, lock = [ctx.chain_init_output(w, h, 1)], threading.Lock()

        def worker(k):
            r = np.random.default_rng(1000 * seed + k)
            try:
                for _ in range(steps):
                    op = int(r.integers(0, 9))
                    with lock:
                        tok = tokens[int(r.integers(0, len(tokens)))]
                    try:
                        if op == 0:
                            ctx.render(cam if r.random() < 0.5 else cam2, 8, int(r.integers(0, 4)), int(r.integers(0, 2)))
                        elif op == 1:
                            new, _ = ctx.render1_chained(cam, 8, w, h, token=tok, consume=r.random() < 0.2, fetch=("r",) if r.random() < 0.3 else ())
                            with lock:
                                tokens.append(new)
                        elif op == 2:
                            ctx.chain_fetch(tok, w, h, "r g b" if r.random() < 0.5 else "sa sctr")
                        elif op == 3:
                            with lock:
                                victim = tokens.pop(int(r.integers(1, len(tokens)))) if len(tokens) > 3 else 0
                            if victim:
                                ctx.chain_release(victim)
                        elif op == 4:
                            ctx.upload_state(*ctx.download_state())
                        elif op == 5:
                            ctx.download_color(); ctx.present(2); ctx.stats(); ctx.chain_info()
                        elif op == 6:
                            ctx.render(cam, -1, 1)                  # refused: a message of this thread's own
                        elif op == 7:
                            ctx.render1(cam, 8, w, h, planes(w, h, k))
                        else:
                            ctx.reseed(int(r.integers(0, 1 << 40)))
                    except B.PtmiError as e:
                        if op == 6 and "bounce_limit" not in str(e):
                            problems.append("thread %d was handed another call's message: %s" % (k, e))
            except Exception as e:                                   # noqa: BLE001
                problems.append(repr(e))

        ts = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not problems, problems[:3]


def group():
    sp, pl = pkg.world.scene16()
    for devices, stripe, (w, h) in (([0, 0, 0], 4, (40, 30)), ([0, 0], 0, (32, 16)), ([0] * 8, 2, (16, 9)), ([0], 8, (24, 24))):
        with pkg.Group(devices, stripe) as g:
            assert g.size == len(devices)
            g.set_scene(sp, pl)
            g.resize(w, h)
            g.init_output(3)
            g.render(cam, 8, 2)
            g.render(cam, 8, 2, pkg.STREAMS)
            g.set_option(B.OPT_SPP_CHUNKS, 2)
            g.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)          # the stream form reads stream lengths back while it runs: a host thread per member
            g.render(cam, 8, 2, pkg.STREAMS)
            g.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
            g.set_variant(0)
            g.reseed(4)
            g.synchronize()
            g.download_color()
            g.stats()
            g.member(0).stats()
            with DeviceBlocks([(w + 8) * h * 4] * 3) as dst:
                try:
                    for root in (0, len(devices) - 1, 0):
                        g.gather_color(root, *dst)
                    g.resize(w + 8, h)                               # another size: the gather's blocks are made anew
                    g.init_output(5)
                    g.render(cam, 8, 1)
                    g.gather_color(0, *dst)
                    if len(devices) > 1:                             # a member resized behind the group's back: the read-outs refuse, nothing is copied
                        for other in ((g.width + 8, g.height + 16), (g.width - 8, g.height)):    # (another height: other rows; another width only: the same rows)
                            g.member(1).resize(*other)
                            for read in (g.download_color, lambda: g.gather_color(0, *dst)):
                                try:
                                    read()
                                    raise AssertionError("a member of another size was read out by the group's size")
                                except B.PtmiError as e:
                                    assert e.code == B.PTMI_ESTATE or not PLAIN, e
                        g.resize(w, h)                               # the group's resize puts every member right again
                        g.init_output(6)
                        g.download_color()
                finally:
                    quiet(g.synchronize)


def refusals():
    """Arguments the ABI refuses: nothing may be touched, the context stays usable."""
    sp, pl = pkg.world.scene16()
    lib = B.load_library()
    assert lib.ptmi_destroy(None) is None
    assert lib.ptmi_render(None, None, 0, 0, 0) != 0
    with pkg.Context(0) as ctx:
        bad = [lambda: ctx.render(cam, 8, 1), lambda: ctx.resize(0, 4), lambda: ctx.resize(4, -1), lambda: ctx.download_color(),
               lambda: ctx.set_scene(sp[:0], pl[:0]), lambda: ctx.set_partition(0, 2, 0), lambda: ctx.set_partition(8, 2, 2),
               lambda: ctx.set_option(99, 1), lambda: ctx.get_option(99), lambda: ctx.set_option(B.OPT_STREAMS_FORM, 7),
               lambda: ctx.set_option(B.OPT_PASS_HANDOFF, 9), lambda: ctx.set_option(B.OPT_CHAIN_SLOTS, -1), lambda: ctx.set_variant(12345),
               lambda: ctx.present(1), lambda: ctx.init_output(1)]
        for call in bad:
            try:
                call()
            except B.PtmiError:
                continue
            raise AssertionError("a refusal was expected")
        many_s = np.concatenate([sp] * 80)                        # 1120 spheres > PTMI_MAX_PRIMITIVES
        try:
            ctx.set_scene(many_s, pl)
            raise AssertionError("too many primitives were accepted")
        except B.PtmiError:
            pass
        ctx.set_scene(sp, pl)
        ctx.resize(16, 16)
        ctx.init_output(1)
        for call in (lambda: ctx.render(cam, -1, 1), lambda: ctx.render(cam, 8, -1), lambda: ctx.render(cam, 8, 1, 5), lambda: ctx.present(0),
                     lambda: ctx.render1_chained(cam, 8, 0, 4), lambda: ctx.chain_init_output(-1, 4, 1)):
            try:
                call()
            except B.PtmiError:
                continue
            raise AssertionError("a refusal was expected")
        ctx.render(cam, 8, 1)
        ctx.download_color()
    for devices in ([], [0, 1], [3]):
        try:
            with pkg.Group(devices, 8):
                pass
            raise AssertionError("a group over devices that do not exist was accepted")
        except B.PtmiError:
            pass


SCENARIOS = [resident, roundtrips, partitioned, glass, stream_overflow, closures, staged, threads, group, refusals]


def nothing_left(where):
    left = (stub.hipstub_live_blocks(), stub.hipstub_live_host_blocks(), stub.hipstub_live_streams(), stub.hipstub_live_events())
    if left[0]:
        stub.hipstub_print_live()
    assert left == (0, 0, 0, 0), "%s: (device blocks, pinned blocks, streams, events) still alive = %r" % (where, left)


def main():
    global PLAIN
    only = os.environ.get("PTMI_HOSTSAN_ONLY")
    if only and only.startswith("monkey"):                          # monkey[_threads]:<first seed>:<seeds>:<steps>
        which, first, count, steps = (only.split(":") + ["0", "8", "1500"])[:4]
        stub.hipstub_set_device_size(8, 4 << 30)
        for seed in range(int(first), int(first) + int(count)):
            (monkey_threads if which == "monkey_threads" else monkey)(seed, int(steps))
            nothing_left("monkey, seed %d" % seed)
            print("hostsan monkey seed %d: %d steps, %d kernel launches so far" % (seed, int(steps), stub.hipstub_launches(b"")), flush=True)
        assert stub.hipstub_stale_errors() == 0
        print("sanitized host side: done")
        return
    stride = int(os.environ.get("PTMI_HOSTSAN_STRIDE", "1"))
    if os.environ.get("PTMI_HOSTSAN_DEFERRED") == "1":              # asynchronous work happens at the next synchronisation that covers it
        stub.hipstub_set_deferred(1)
    stub.hipstub_set_device_size(8, 4 << 30)
    report = {}
    for sc in SCENARIOS:
        if only and sc.__name__ not in only.split(","):
            continue
        before = [stub.hipstub_calls(k) for k in range(6)]
        launches0 = stub.hipstub_launches(b"")
        PLAIN = True
        sc()
        PLAIN = False
        nothing_left(sc.__name__)
        counts = [stub.hipstub_calls(k) - before[k] for k in range(6)]
        launches = stub.hipstub_launches(b"") - launches0
        walked = absorbed = 0
        # one failure; a failure and the next call of the kind failing too (the recovery's own call); a device that stays broken
        for kind, n, more in [(kind, n, more) for more in MORE for kind, n in enumerate(counts)]:
            for k in range(1, n + 1, stride if more == 0 else stride * MORE_STRIDE):
                stub.hipstub_fail_run(kind, k, more)
                try:
                    sc()
                    absorbed += 1                                   # the library made do without (a fallback), or the failure fell into an expected refusal
                    if os.environ.get("HIPSTUB_TRACE"):
                        print("ABSORBED: %s, the %d-th %s" % (sc.__name__, k, KINDS[kind]), file=sys.stderr, flush=True)
                except (B.PtmiError, MemoryError):
                    pass
                finally:
                    stub.hipstub_fail(kind, 0)
                    left_behind = stub.hipstub_clear_error()
                walked += 1
                # an error the library reported (or chose to ignore) is not left in the runtime's sticky slot for the next launch check to find
                assert left_behind == 0, "%s with the %d-th %s failing: hipError %d was left in the sticky slot" % (sc.__name__, k, KINDS[kind], left_behind)
                nothing_left("%s with the %d-th %s failing (and %d more)" % (sc.__name__, k, KINDS[kind], more))
        report[sc.__name__] = {"calls": dict(zip(KINDS, counts)), "kernel_launches": launches, "failure_points_walked": walked, "absorbed": absorbed,
                               "stale_errors_handed_to_a_launch": stub.hipstub_stale_errors()}
        print("hostsan %s: %r" % (sc.__name__, report[sc.__name__]), flush=True)
    # ... and no launch of the library was ever blamed for an older call's error (hipGetLastError after a launch hands out the sticky slot)
    assert stub.hipstub_stale_errors() == 0, "%d launches were handed an older call's error" % stub.hipstub_stale_errors()
    print("sanitized host side: done")


if __name__ == "__main__":
    main()
