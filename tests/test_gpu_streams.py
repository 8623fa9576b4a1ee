"""render Streams (src/Scene/Trace.hs:141-191, 272-331) on the device against the oracle's literal
per-pixel restatement: colour adds for every hit, no bounce limit, pixel seed advanced by one draw."""
import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu
CAP = 1 << 16      # kStreamsHardCap; the reference has no cap (Trace.hs:166-170)


@pytest.mark.parametrize("w,h,spp,scene_name", [(96, 64, 1, "main"), (120, 67, 3, "s16"), (33, 5, 2, "main")])
def test_render_streams_matches_oracle(ctx, pkg, ora, w, h, spp, scene_name):
    scene = pkg.world.main_scene() if scene_name == "main" else pkg.world.scene16()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, w, h)
    ctx.set_scene(*scene)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.reset_stats()
    ctx.render(cam, 15, spp, pkg.STREAMS)
    got = ctx.download_state()
    want, live = ora.render_streams(scene[0], scene[1], cam, w, h, CAP, spp, start)
    assert_planes_equal(got, want, "render Streams %dx%d spp %d" % (w, h, spp))
    st = ctx.stats()
    assert st["live_bounces"] == live
    assert 1 <= st["stream_iterations"] < 64


def test_streams_ignores_the_iteration_limit_like_the_reference(ctx, pkg, ora):
    """notFinished never stops a non-empty stream (Trace.hs:166-170): maxIterations has no effect."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 64, 40
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    out = []
    for limit in (1, 15):
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.render(cam, limit, 2, pkg.STREAMS)
        out.append(ctx.download_state())
    assert_planes_equal(out[0], out[1], "limit 1 vs 15")


def test_streams_seed_rule_and_first_sample_colour(ctx, pkg, ora):
    """Sample 1: Streams == Inline in colour except where a path outlives the limit or ends on a hit with
    near-zero throughput; carried seed = original advanced by ONE draw (updateSeed)."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 80, 60
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.render(cam, 15, 1, pkg.STREAMS)
    stm = ctx.download_state()
    ctx.upload_state(*start)
    ctx.render(cam, 64, 1, pkg.INLINE)
    inl = ctx.download_state()
    same = np.mean((stm[0] == inl[0]) & (stm[1] == inl[1]) & (stm[2] == inl[2]))
    assert same > 0.98
    for i in (0, 17, w * h - 1):
        _, _, end = ora.sfc32_stream([p.reshape(-1)[i] for p in start[3:]], 1)
        assert tuple(int(p.reshape(-1)[i]) for p in stm[3:]) == tuple(end)


def test_render1_streams(ctx, pkg, ora):
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 50, 30
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    got = ctx.render1(cam, 15, w, h, start, algorithm=pkg.STREAMS)
    want, _ = ora.render_streams(sp, pl, cam, w, h, CAP, 1, start)
    assert_planes_equal(got, want, "render1 Streams")
