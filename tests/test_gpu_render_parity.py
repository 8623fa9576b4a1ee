"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bar: BIT-EXACT on all seven planes (stricter than north_star's 1e-4
relative on the RGB accumulator; the stated tolerance is also checked, trivially)."""
import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4   # north_star: "within 1e-4 relative on the RGB accumulator"


def run_gpu(ctx, pkg, scene, cam, w, h, limit, spp, start, algorithm=None):
    ctx.set_scene(*scene)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.reset_stats()
    ctx.render(cam, limit, spp, pkg.INLINE if algorithm is None else algorithm)
    return ctx.download_state(), ctx.stats()


@pytest.mark.parametrize("w,h,limit,spp,scene_name", [
    (256, 256, 4, 1, "main"),      # BASELINE.json configs[0]
    (200, 150, 15, 1, "main"),     # reference-native bounce limit (Trace.hs:200), reduced size
    (160, 90, 8, 4, "s16"),        # configs[1] shape, reduced
    (97, 61, 15, 3, "s16"),        # ragged: not a multiple of the 256-lane workgroup
    (64, 1, 8, 2, "main"),
    (1, 64, 8, 2, "main"),
    (48, 40, 8, 700, "s16"),       # many restarts per lane: long per-pixel chains of samples
    (32, 24, 64, 33, "main"),      # a bounce limit no path reaches
])
def test_render_inline_matches_oracle(ctx, pkg, ora, w, h, limit, spp, scene_name):
    scene = pkg.world.main_scene() if scene_name == "main" else pkg.world.scene16()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, w, h)
    got, stats = run_gpu(ctx, pkg, scene, cam, w, h, limit, spp, start)
    want, live = ora.render_inline(scene[0], scene[1], cam, w, h, limit, spp, start)
    assert_planes_equal(got, want, "render Inline %dx%d limit %d spp %d" % (w, h, limit, spp))
    for a, b in zip(got[:3], want[:3]):
        assert np.all(np.abs(a - b) <= REL_TOL * np.abs(b))
    assert stats["live_bounces"] == live
    assert stats["nominal_bounces"] == w * h * spp * limit


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18])
def test_every_kernel_variant_matches_oracle(pkg, ora, ablations, variant):
    """All loop shapes (persistent hand-out, lock step, regenerate, cached/static, LDS or scalar-load scene,
    second shade round pooled over 2 / 4 / 8 waves, 8x8 / 16x4 / 4x16 / 32x2 pixel tiles per wave,
    round 1's [shade][shade][trace] loop without the frozen-shade shortcut)
    compute the same seven planes -- they differ only in how lanes are kept busy."""
    scene = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h, limit, spp = 150, 90, 8, 5
    start = initial_planes(ora, w, h)
    want, live = ora.render_inline(scene[0], scene[1], cam, w, h, limit, spp, start)
    with pkg.Context(0, library=ablations) as c:            # the ablation kernels live in libptmi_ablations.so only
        c.set_variant(variant)
        got, stats = run_gpu(c, pkg, scene, cam, w, h, limit, spp, start)
    assert_planes_equal(got, want, "variant %d" % variant)
    assert stats["live_bounces"] == live
    with pkg.Context(0) as c:                               # the product library: what `auto` chooses from, nothing else
        if variant in (4, 5, 13, 17):
            c.set_variant(variant)
            got, stats = run_gpu(c, pkg, scene, cam, w, h, limit, spp, start)
            assert_planes_equal(got, want, "variant %d, product library" % variant)
        else:
            with pytest.raises(pkg.PtmiError) as e:
                c.set_variant(variant)
            assert e.value.code == -1


def test_non_finite_inputs_take_the_literal_fold(ctx, pkg, ora):
    """A camera at infinity makes every key NaN / inf: check_hit must fall back to the literal fold and
    still agree with the oracle on which pixels count as hits (colour planes: NaN == NaN by position)."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.camera((np.inf, 0.0, 0.0), (0.0, 0.0, 0.0), 90)
    w, h = 32, 16
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.render(cam, 4, 1)
    got = ctx.download_state()
    with np.errstate(all="ignore"):
        want, _ = ora.render_inline(sp, pl, cam, w, h, 4, 1, start)
    for a, b in zip(got[:3], want[:3]):
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)          # the RNG planes carry no floating point


def test_cost_ordered_dispatch_changes_nothing_but_the_order(pkg, ora):
    """The tiled kernels record what every quad of tiles costs in the first launch with a given camera, sort the quads
    on the device in the second and dispatch the most expensive first from then on.  Planes must not
    notice: five launches with one camera, a camera change, back again, a scene change, both algorithms."""
    sp, pl = pkg.world.scene16()
    cam1 = pkg.world.initial_camera()
    cam2 = pkg.world.camera((1.0, 2.0, 3.0), (0.1, -0.2, 0.4), 70)
    w, h, limit = 200, 120, 8
    for algorithm, ref in ((pkg.INLINE, ora.render_inline), (pkg.STREAMS, None)):
        start = initial_planes(ora, w, h)
        with pkg.Context(0) as c:
            c.set_scene(sp, pl)
            c.resize(w, h)
            c.upload_state(*start)
            want = start
            schedule = [(cam1, 2), (cam1, 1), (cam1, 3), (cam1, 1), (cam1, 2), (cam2, 2), (cam2, 2), (cam2, 1), (cam1, 1), (cam1, 2)]
            for k, (cam, spp) in enumerate(schedule):
                c.render(cam, limit, spp, algorithm)
                if algorithm == pkg.INLINE:
                    want, _ = ora.render_inline(sp, pl, cam, w, h, limit, spp, want)
                else:
                    want, _ = ora.render_streams(sp, pl, cam, w, h, 1 << 16, spp, want)
                assert_planes_equal(c.download_state(), want, "launch %d" % k)
            sp2, pl2 = pkg.world.main_scene()
            c.set_scene(sp2, pl2)
            for k in range(3):
                c.render(cam1, limit, 2, algorithm)
                if algorithm == pkg.INLINE:
                    want, _ = ora.render_inline(sp2, pl2, cam1, w, h, limit, 2, want)
                else:
                    want, _ = ora.render_streams(sp2, pl2, cam1, w, h, 1 << 16, 2, want)
                assert_planes_equal(c.download_state(), want, "after the scene change, launch %d" % k)


@pytest.mark.parametrize("w,h,spp,chunks", [(96, 64, 7, 3), (333, 131, 12, 5), (200, 77, 64, 64), (64, 16, 5, 2), (1920, 1080, 9, 4)])
def test_sample_chunks_change_no_bit(pkg, ora, w, h, spp, chunks):
    """PTMI_OPT_SPP_CHUNKS: the tile grid is launched `chunks` times over, copy c rendering a slice of the samples
    after copy c-1 of the same tile has published its planes.  On a small image all copies are resident at once, so
    every later copy really waits for its predecessor -- the result must equal the oracle (and the unchunked launch)."""
    B = pkg.binding
    scene = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, w, h)
    with pkg.Context(0) as c:
        c.set_option(B.OPT_SPP_CHUNKS, chunks)
        assert c.get_option(B.OPT_SPP_CHUNKS) == chunks
        got, stats = run_gpu(c, pkg, scene, cam, w, h, 8, spp, start)
        c.set_option(B.OPT_SPP_CHUNKS, 1)
        plain, _ = run_gpu(c, pkg, scene, cam, w, h, 8, spp, start)
    assert_planes_equal(got, plain, "chunked vs one launch")
    if w * h <= 100000:
        want, live = ora.render_inline(scene[0], scene[1], cam, w, h, 8, spp, start)
        assert_planes_equal(got, want, "%d sample chunks" % chunks)
        assert stats["live_bounces"] == live
