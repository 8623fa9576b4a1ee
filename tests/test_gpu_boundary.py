"""The drop-in boundary on the device: compat entry (= the closure compileFor builds, app/Main.hs:188-191),
resident entry, state programs (initialOutput / reseed / createWith), partitions, error behaviour."""
import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fresh(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def test_render1_is_one_application_of_render(ctx, pkg, ora):
    """host planes in -> new host planes out, one sample; inputs are not modified."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 100, 75
    start = initial_planes(ora, w, h)
    keep = [a.copy() for a in start]
    ctx.set_scene(sp, pl)
    out1 = ctx.render1(cam, 15, w, h, start)
    assert_planes_equal(start, keep, "inputs untouched")
    want1, _ = ora.render_inline(sp, pl, cam, w, h, 15, 1, start)
    assert_planes_equal(out1, want1, "render1 #1")
    out2 = ctx.render1(cam, 15, w, h, out1)                       # (it + 1, dewit (scalar c) acc)
    want2, _ = ora.render_inline(sp, pl, cam, w, h, 15, 2, start)
    assert_planes_equal(out2, want2, "render1 #2")


def test_render1_with_explicit_screen_pixels(ctx, pkg, ora):
    """The Matrix (V2 Int) argument (Util.hs:209-210): screenPixels gives the implicit result; a flipped
    matrix renders the mirrored image with the un-mirrored seeds (generality of render's first argument)."""
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h = 64, 40
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    sx, sy = pkg.world.screen_pixels(w, h)
    implicit = ctx.render1(cam, 8, w, h, start)
    explicit = ctx.render1(cam, 8, w, h, start, screen=(sx, sy))
    assert_planes_equal(explicit, implicit, "screenPixels")
    flipped = (np.ascontiguousarray(sx[:, ::-1]), sy)
    got = ctx.render1(cam, 8, w, h, start, screen=flipped)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 1, start, screen=flipped)
    assert_planes_equal(got, want, "flipped screen")


def test_resident_render_equals_repeated_render1(ctx, pkg, ora):
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 80, 50
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.render(cam, 15, 3)
    resident = ctx.download_state()
    acc = start
    for _ in range(3):
        acc = ctx.render1(cam, 15, w, h, acc)
    assert_planes_equal(resident, acc, "render(n_spp=3) vs 3 x render1")
    # and split launches compose: 1 + 2
    ctx.upload_state(*start)
    ctx.render(cam, 15, 1)
    ctx.render(cam, 15, 2)
    assert_planes_equal(ctx.download_state(), acc, "1 + 2 spp")


def test_init_output_reseed_create_with(ctx, pkg, ora):
    """initialOutput / reseed (Util.hs:134-135, :204-205) and createWith (Util.hs:125) on the device."""
    w, h = 70, 33
    ctx.set_scene(*pkg.world.main_scene())
    ctx.resize(w, h)
    ctx.init_output(0xABCDEF0123)
    got = ctx.download_state()
    seeds = ora.gen_seeds(0xABCDEF0123, 0, w * h)
    assert all(np.all(p == 0) for p in got[:3])
    for a, b in zip(got[3:], seeds):
        assert np.array_equal(a.reshape(-1), b)
    ctx.render(pkg.world.initial_camera(), 4, 1)
    col = ctx.download_color()
    ctx.reseed(77)
    got = ctx.download_state()
    assert_planes_equal(got[:3], col, "reseed keeps colour")
    for a, b in zip(got[3:], ora.gen_seeds(77, 0, w * h)):
        assert np.array_equal(a.reshape(-1), b)
    r = np.random.default_rng(3)
    words = [r.integers(0, 2 ** 32, w * h, dtype=np.uint32) for _ in range(3)]
    ctx.create_with(*words)
    got = ctx.download_state()
    for i in (0, 1, 1234, w * h - 1):
        assert tuple(int(p.reshape(-1)[i]) for p in got[3:]) == ora.sfc32_seed3(words[0][i], words[1][i], words[2][i])


@pytest.mark.parametrize("n_parts,stripe,h", [(2, 8, 64), (3, 4, 50), (8, 8, 61)])
def test_row_stripe_partitions_reproduce_the_whole_image(pkg, ora, n_parts, stripe, h):
    """Every part renders only its rows, from seeds of the GLOBAL pixel index; stitched = unpartitioned."""
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w = 48
    start = initial_planes(ora, w, h)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 2, start)
    stitched = [np.zeros_like(p) for p in want]
    for part in range(n_parts):
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.set_partition(stripe, n_parts, part)
            c.resize(w, h)
            rows = c.global_rows()
            c.init_output(0x5EED1234)
            seeded = c.download_state()
            for a, b in zip(seeded[3:], start[3:]):
                assert np.array_equal(a, b[rows])
            c.render(cam, 8, 2)
            for dst, src in zip(stitched, c.download_state()):
                dst[rows] = src
    assert_planes_equal(stitched, want, "%d stripes" % n_parts)


def test_edge_sizes_and_degenerate_counts(ctx, pkg, ora):
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    ctx.set_scene(sp, pl)
    for (w, h, limit, spp) in [(1, 1, 15, 1), (3, 2, 0, 4), (5, 7, 15, 0), (257, 3, 1, 1)]:
        start = initial_planes(ora, w, h)
        start[1][:] = 0.25
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.render(cam, limit, spp)
        want, _ = ora.render_inline(sp, pl, cam, w, h, limit, spp, start)
        assert_planes_equal(ctx.download_state(), want, str((w, h, limit, spp)))


def test_single_primitive_and_planes_only_scenes(ctx, pkg, ora):
    """expMinWith's one-element case (Util.hs:173) and a scene without spheres."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 40, 30
    start = initial_planes(ora, w, h)
    for s, p in [(sp[:1], pl[:0]), (sp[:0], pl), (sp[:0], pl[:1]), (sp[3:5], pl[:0])]:
        ctx.set_scene(s, p)
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.render(cam, 6, 2)
        want, _ = ora.render_inline(s, p, cam, w, h, 6, 2, start)
        assert_planes_equal(ctx.download_state(), want, "scene %d+%d" % (len(s), len(p)))


def test_other_cameras(ctx, pkg, ora):
    sp, pl = pkg.world.scene16()
    ctx.set_scene(sp, pl)
    w, h = 64, 36
    start = initial_planes(ora, w, h)
    for cam in [pkg.world.camera((0, 0, 0), (0, 0, 0), 90), pkg.world.camera((3, 1, 2), (1.0, 2.5, -0.7), 60),
                pkg.world.camera((1, -1.6, -4.8), (250.0, -130.0, 999.0), 120)]:       # sin/cos beyond |x| = 120
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.render(cam, 8, 1)
        want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 1, start)
        assert_planes_equal(ctx.download_state(), want, "camera %s" % cam)


def test_error_behaviour(fresh, pkg):
    """0 / negative codes + message, nothing throws across the ABI, no abort (SURVEY.md 8b 'Errors')."""
    c, cam = fresh, pkg.world.initial_camera()
    sp, pl = pkg.world.main_scene()
    with pytest.raises(pkg.PtmiError) as e:
        c.render(cam, 8, 1)
    assert e.value.code == -5                      # PTMI_ESTATE: no scene
    with pytest.raises(pkg.PtmiError) as e:
        c.set_scene(sp[:0], pl[:0])
    assert e.value.code == -1                      # empty scene = expMinWith [] (Util.hs:172)
    c.set_scene(sp, pl)
    with pytest.raises(pkg.PtmiError) as e:
        c.render(cam, 8, 1)
    assert e.value.code == -5                      # no resize yet
    with pytest.raises(pkg.PtmiError) as e:
        c.resize(0, 10)
    assert e.value.code == -1
    c.resize(8, 8)
    with pytest.raises(pkg.PtmiError) as e:
        c.render(cam, -1, 1)
    assert e.value.code == -1
    bad = sp.copy()
    bad["brdf_tag"][0] = 7
    with pytest.raises(pkg.PtmiError) as e:
        c.set_scene(bad, pl)
    assert e.value.code == -1
    with pytest.raises(pkg.PtmiError) as e:
        pkg.Context(10 ** 6)
    assert e.value.code == -2
    c.render(cam, 8, 1)                            # still usable after errors
    c.synchronize()


def test_another_callers_runtime_error_is_neither_tripped_over_nor_swallowed(fresh, pkg, ora):
    """hipGetLastError() keeps the thread's last failure until somebody asks.  A failed call of the application's (here: an allocation of
    2^60 bytes) must not make this library's next launches report failure -- they return hipLaunchKernel's own status
    (csrc/ptmi_kernels.h: launch) -- and the error must still be there for its owner afterwards."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h = 64, 32
    fresh.set_scene(sp, pl)
    fresh.resize(w, h)
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 60))
    assert rc != 0 and not p.value                                  # the application's own failure, left in the sticky slot
    try:
        fresh.init_output(77)                                       # seed kernel
        fresh.render(cam, 8, 2)                                     # render Inline
        fresh.render(cam, 8, 1, pkg.STREAMS)
        got = fresh.download_state()
        out1 = fresh.render1(cam, 8, w, h, got)                     # the closure, copying
        tok, _ = fresh.render1_chained(cam, 8, w, h, planes_in=got) # ... and chained
        out2 = fresh.chain_fetch(tok, w, h)
        fresh.present(3)
        assert hip.hipGetLastError() == rc                          # still the application's to find
    finally:
        hip.hipGetLastError()
    start = initial_planes(ora, w, h, seed0=77)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 2, start)
    want, _ = ora.render_streams(sp, pl, cam, w, h, 1 << 10, 1, want)
    assert_planes_equal(got, want, "renders issued with a foreign error pending")
    assert_planes_equal(out1, out2, "the two closures")


def test_a_resize_the_device_cannot_hold_leaves_the_context_unsized_and_usable(fresh, pkg, ora):
    """ptmi_resize frees the old planes before it asks for the new ones.  When the new ones cannot be had (here: 2^35 pixels, ~960 GB) the
    context must not keep pointing into the block it freed: it is UNSIZED (PTMI_ESTATE) until a resize succeeds, and then exact.  Found by
    the random walk of tests/test_host_sanitized.py on the HIP stand-in; this is the same path against the real runtime's out-of-memory."""
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    fresh.set_scene(sp, pl)
    fresh.resize(64, 32)
    fresh.init_output(5)
    fresh.render(cam, 8, 1)
    with pytest.raises(pkg.PtmiError) as err:
        fresh.resize(1 << 16, 1 << 19)
    assert err.value.code == pkg.binding.PTMI_ENOMEM, err.value
    with pytest.raises(pkg.PtmiError) as err:
        fresh.resize(2 ** 31 - 1, 2 ** 31 - 1)                      # seven planes of 2^62 pixels would wrap a size_t: refused by count
    assert err.value.code == pkg.binding.PTMI_ELIMIT, err.value
    for call in (fresh.download_color, lambda: fresh.render(cam, 8, 1), lambda: fresh.init_output(1), lambda: fresh.present(1)):
        with pytest.raises(pkg.PtmiError) as err:
            call()
        assert err.value.code == pkg.binding.PTMI_ESTATE, err.value
    w, h = 96, 40
    fresh.resize(w, h)
    fresh.init_output(0x5EED1234)
    fresh.render(cam, 8, 2)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 2, initial_planes(ora, w, h))
    assert_planes_equal(fresh.download_state(), want, "after a failed resize")
    with pkg.Group([0, 0], 8) as g:                                 # the group: unsized as a whole when one member's resize fails
        g.set_scene(sp, pl)
        g.resize(w, h)
        with pytest.raises(pkg.PtmiError):
            g.resize(1 << 16, 1 << 20)
        with pytest.raises(pkg.PtmiError) as err:
            g.download_color()
        assert err.value.code == pkg.binding.PTMI_ESTATE, err.value
        g.resize(w, h)
        g.init_output(0x5EED1234)
        g.render(cam, 8, 2)
        got = g.download_color()
        for k in range(3):
            assert np.array_equal(got[k].view(np.uint32), np.asarray(want[k]).view(np.uint32))
        g.member(1).resize(w - 8, h)                                # a member sized behind the group's back (same rows, another width)
        with pytest.raises(pkg.PtmiError) as err:
            g.download_color()                                      # ... is not copied by the group's size
        assert err.value.code == pkg.binding.PTMI_ESTATE, err.value


def test_context_is_usable_from_other_threads(ctx, pkg, ora):
    """The closure may be forced on any of three OS threads (app/Main.hs:178-180)."""
    import threading
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 32, 32
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 1, start)
    results = {}

    def work(i):
        results[i] = ctx.render1(cam, 8, w, h, start)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(4):
        assert_planes_equal(results[i], want, "thread %d" % i)


@pytest.mark.parametrize("w,h", [(64, 48), (61, 7), (1, 1)])
def test_present_matches_graphics_loop_arithmetic(ctx, pkg, ora, w, h):
    """app/Main.hs:351 (zipWith3 V3 r g b) + fs.glsl:12 (texture.rgb / u_iterations) + unorm8 framebuffer."""
    sp, pl = pkg.world.main_scene()
    ctx.set_scene(sp, pl)
    ctx.resize(w, h)
    ctx.init_output(5)
    iters = 7
    ctx.render(pkg.world.initial_camera(), 15, iters)
    r, g, b = ctx.download_color()
    rgb, rgba = ctx.present(iters)
    want = np.stack([r, g, b], -1) / np.float32(iters)
    assert np.array_equal(rgb.view(np.uint32), want.astype(np.float32).view(np.uint32))
    with np.errstate(invalid="ignore"):
        clamped = np.where(want > 0, np.minimum(want, np.float32(1.0)), np.float32(0.0)).astype(np.float32)
    want8 = (clamped * np.float32(255.0) + np.float32(0.5)).astype(np.uint32).astype(np.uint8)
    assert np.array_equal(rgba[..., :3], want8) and np.all(rgba[..., 3] == 255)
    with pytest.raises(pkg.PtmiError):
        ctx.present(0)


@pytest.mark.parametrize("n_spheres,n_planes", [(200, 3), (1000, 24)])
def test_large_scenes_up_to_the_primitive_limit(ctx, actx, pkg, ora, n_spheres, n_planes):
    """PTMI_MAX_PRIMITIVES = 1024.  Big scenes are read through scalar loads instead of LDS (occupancy);
    every variant must still agree with the oracle.  One more primitive is refused with PTMI_ELIMIT."""
    r = np.random.default_rng(n_spheres)
    w = pkg.world
    spheres = np.zeros(n_spheres, w.SPHERE_DTYPE)
    spheres["position"] = r.uniform(-30, 30, (n_spheres, 3))
    spheres["radius"] = r.uniform(0.2, 1.5, n_spheres)
    spheres["color"] = r.uniform(0.1, 1, (n_spheres, 3))
    spheres["illuminance"] = np.where(r.random(n_spheres) < 0.1, 50.0, 0.0)
    spheres["brdf_tag"] = r.integers(0, 2, n_spheres)
    spheres["brdf_param"] = r.uniform(0.2, 1.0, n_spheres)
    planes = np.zeros(n_planes, w.PLANE_DTYPE)
    planes["position"] = r.uniform(-40, 40, (n_planes, 3))
    nrm = r.normal(0, 1, (n_planes, 3))
    planes["direction"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    planes["color"] = r.uniform(0.2, 1, (n_planes, 3))
    planes["brdf_tag"] = r.integers(0, 2, n_planes)
    planes["brdf_param"] = 0.8
    cam = w.initial_camera()
    wd, ht = 64, 40
    start = initial_planes(ora, wd, ht)
    want, _ = ora.render_inline(spheres, planes, cam, wd, ht, 6, 2, start)
    for variant in (0, 4):
        ctx.set_variant(variant)
        ctx.set_scene(spheres, planes)
        ctx.resize(wd, ht)
        ctx.upload_state(*start)
        ctx.render(cam, 6, 2)
        assert_planes_equal(ctx.download_state(), want, "%d+%d primitives, variant %d" % (n_spheres, n_planes, variant))
    ctx.set_variant(0)
    ctx.upload_state(*start)
    ctx.render(cam, 6, 1, pkg.STREAMS)
    want_s, _ = ora.render_streams(spheres, planes, cam, wd, ht, 1 << 16, 1, start)
    assert_planes_equal(ctx.download_state(), want_s, "streams, %d primitives" % (n_spheres + n_planes))
    # the paths that used to stage the whole scene in LDS whatever its size: the stream form of Streams, the
    # degenerate-count route (n_spp = 0 / limit = 0) and the lock-step / regenerate / pooled variants
    ctx.set_option(pkg.binding.OPT_STREAMS_FORM, pkg.binding.FORM_STREAM)
    ctx.upload_state(*start)
    ctx.render(cam, 6, 1, pkg.STREAMS)
    assert_planes_equal(ctx.download_state(), want_s, "streams (stream form), %d primitives" % (n_spheres + n_planes))
    ctx.set_option(pkg.binding.OPT_STREAMS_FORM, pkg.binding.FORM_AUTO)
    for limit, spp in ((6, 0), (0, 2)):
        ctx.upload_state(*start)
        ctx.render(cam, limit, spp)
        want_d, _ = ora.render_inline(spheres, planes, cam, wd, ht, limit, spp, start)
        assert_planes_equal(ctx.download_state(), want_d, "limit %d, spp %d, %d primitives" % (limit, spp, n_spheres + n_planes))
    actx.set_scene(spheres, planes)                          # the ablation library's persistent / lock-step / regenerate / pooled kernels
    actx.resize(wd, ht)
    for variant in (1, 2, 3, 10):
        actx.set_variant(variant)
        actx.upload_state(*start)
        actx.render(cam, 6, 2)
        assert_planes_equal(actx.download_state(), want, "%d+%d primitives, variant %d" % (n_spheres, n_planes, variant))
    actx.set_variant(0)
    if n_spheres + n_planes == 1024:
        with pytest.raises(pkg.PtmiError) as e:
            ctx.set_scene(np.concatenate([spheres, spheres[:1]]), planes)
        assert e.value.code == -6


def test_staged_host_transfers_equal_plain_copies(pkg, ora, monkeypatch):
    """Host-buffer entry points move large planes through the pinned ring with worker threads (ptmi_stage.h);
    PTMI_STAGE_THREADS=0 switches the engine off.  Same bytes either way, at a size whose planes do not divide
    into whole chunks or pieces, with and without the two Int64 screen planes, and equal to the oracle."""
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h = 2203, 1001                                           # 8.8 MB per plane: an 8-MB chunk and a ragged one; Int64 planes: 3 chunks
    start = initial_planes(ora, w, h)
    rng = np.random.default_rng(7)
    start = tuple(rng.random((h, w), dtype=np.float32) for _ in range(3)) + tuple(start[3:])
    sx, sy = pkg.world.screen_pixels(w, h)
    flipped = (np.ascontiguousarray(sx[:, ::-1]), sy)
    results = {}
    for threads in ("0", "3", None):
        if threads is None:
            monkeypatch.delenv("PTMI_STAGE_THREADS", raising=False)
        else:
            monkeypatch.setenv("PTMI_STAGE_THREADS", threads)
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            one = c.render1(cam, 8, w, h, start)
            two = c.render1(cam, 8, w, h, one, screen=flipped)
            c.resize(w, h)
            c.upload_state(*two)
            back = c.download_state()
            assert_planes_equal(back, two, "upload/download round trip, threads=%s" % threads)
            c.render(cam, 8, 1)
            three = c.download_state()
            rgb, rgba = c.present(3)
        results[threads] = (one, two, three, rgb, rgba)
    want1, _ = ora.render_inline(sp, pl, cam, w, h, 8, 1, start)
    want2, _ = ora.render_inline(sp, pl, cam, w, h, 8, 1, want1, screen=flipped)
    want3, _ = ora.render_inline(sp, pl, cam, w, h, 8, 1, want2)
    for threads, (one, two, three, rgb, rgba) in results.items():
        assert_planes_equal(one, want1, "render1, threads=%s" % threads)
        assert_planes_equal(two, want2, "render1 with screen planes, threads=%s" % threads)
        assert_planes_equal(three, want3, "resident after upload, threads=%s" % threads)
        assert np.array_equal(rgb, results["0"][3]) and np.array_equal(rgba, results["0"][4])


def test_contracted_arithmetic_is_a_labelled_mode_and_never_the_default(pkg, ora):
    """PTMI_OPT_ARITHMETIC: the default is the literal arithmetic (bit-identical to the oracle); PTMI_ARITH_CONTRACTED -- the same
    kernel with a * b + c fused, a measurement mode -- stays close (most pixels within north_star's 1e-4 after one sample) but is
    NOT the oracle's result; switching back restores it bit for bit."""
    B = pkg.binding
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 320, 200
    start = initial_planes(ora, w, h)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 15, 1, start)
    with pkg.Context(0) as c:
        assert c.get_option(B.OPT_ARITHMETIC) == B.ARITH_EXACT
        c.set_scene(sp, pl)
        c.resize(w, h)
        c.upload_state(*start)
        c.render(cam, 15, 1)
        assert_planes_equal(c.download_state(), want, "default arithmetic")
        c.set_option(B.OPT_ARITHMETIC, B.ARITH_CONTRACTED)
        c.upload_state(*start)
        c.render(cam, 15, 1)
        got = c.download_state()
        close = np.ones((h, w), bool)
        for k in range(3):
            close &= np.abs(got[k] - want[k]) <= 1e-4 * np.abs(want[k])
        assert 0.99 < close.mean()
        assert any(not np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in range(3))     # it really is other arithmetic
        c.set_option(B.OPT_ARITHMETIC, B.ARITH_EXACT)
        c.upload_state(*start)
        c.render(cam, 15, 1)
        assert_planes_equal(c.download_state(), want, "back to the default arithmetic")
        with pytest.raises(pkg.PtmiError):
            c.set_option(B.OPT_ARITHMETIC, 2)
