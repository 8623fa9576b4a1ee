"""render Streams (src/Scene/Trace.hs:141-191, 272-331) on the device against the oracle's literal
per-pixel restatement: colour adds for every hit, no bounce limit, pixel seed advanced by one draw."""
import os

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
    near-zero throughput; under PTMI_SEED_KEEP_ACCUMULATOR the carried seed = original advanced by ONE draw (updateSeed)."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 80, 60
    start = initial_planes(ora, w, h)
    B = pkg.binding
    ctx.set_scene(sp, pl)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_KEEP_ACCUMULATOR)
    try:
        ctx.render(cam, 15, 1, pkg.STREAMS)
        stm = ctx.download_state()
    finally:
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_AUTO)
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


@pytest.mark.parametrize("rule", ["keep", "from_result", "auto"])
@pytest.mark.parametrize("stream_form", [False, True])
@pytest.mark.parametrize("scene_name,w,h,spp", [("main", 96, 64, 3), ("s16", 120, 67, 2)])
def test_both_seed_rules_in_both_forms(ctx, pkg, ora, scene_name, w, h, spp, stream_form, rule):
    """Assumption A5's two readings of `combine` (Trace.hs:179-184) on the device, both forms, against the oracle -- bit for
    bit, seeds included.  PTMI_SEED_AUTO (the default) is `combine new old` = PTMI_SEED_FROM_RESULT for these scenes."""
    B = pkg.binding
    scene = pkg.world.main_scene() if scene_name == "main" else pkg.world.scene16()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, w, h)
    ctx.set_scene(*scene)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.reset_stats()
    assert ctx.get_option(B.OPT_STREAMS_SEED_RULE) == B.SEED_AUTO             # the default
    value = {"keep": B.SEED_KEEP_ACCUMULATOR, "from_result": B.SEED_FROM_RESULT, "auto": B.SEED_AUTO}[rule]
    ctx.set_option(B.OPT_STREAMS_SEED_RULE, value)
    ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if stream_form else B.FORM_AUTO)
    try:
        assert ctx.get_option(B.OPT_STREAMS_SEED_RULE) == value
        ctx.render(cam, 15, spp, pkg.STREAMS)
        got = ctx.download_state()
        live_gpu = ctx.stats()["live_bounces"]
    finally:
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_AUTO)
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
    keep, live_keep = ora.render_streams(scene[0], scene[1], cam, w, h, CAP, spp, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)
    res, live_res = ora.render_streams(scene[0], scene[1], cam, w, h, CAP, spp, start, seed_rule=ora.SEED_FROM_RESULT)
    want, live = (keep, live_keep) if rule == "keep" else (res, live_res)
    assert_planes_equal(got, want, "seed rule %s, %s" % (rule, "stream form" if stream_form else "per-pixel form"))
    assert live_gpu == live
    assert not np.array_equal(keep[3], res[3])               # the two rules really differ


def test_seed_from_result_is_refused_when_rays_split(ctx, pkg):
    B = pkg.binding
    ctx.set_scene(*pkg.world.glass_scene())
    ctx.resize(16, 16)
    ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_FROM_RESULT)
    try:
        with pytest.raises(pkg.PtmiError) as e:
            ctx.render(pkg.world.initial_camera(), 8, 1, pkg.STREAMS)
        assert e.value.code == -1
    finally:
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_AUTO)
    ctx.render(pkg.world.initial_camera(), 8, 1, pkg.STREAMS)         # PTMI_SEED_AUTO: the accumulator's seed with GLASS
    ctx.synchronize()
    with pytest.raises(pkg.PtmiError):
        ctx.set_option(99, 0)
    with pytest.raises(pkg.PtmiError):
        ctx.set_option(B.OPT_STREAM_STEP_CAP, 0)


@pytest.mark.parametrize("stream_form", [False, True])
def test_long_lineages_and_the_step_cap(ctx, pkg, ora, stream_form):
    """An enclosed mirror box: every lineage takes hundreds of traceSteps (the reference has no bound).  With the
    default cap (65536, both forms) nothing is cut and the planes equal the oracle; a small cap cuts the same rays in
    both forms and in the oracle, and the cut rays are COUNTED (ptmi_stats.stream_rays_truncated)."""
    B = pkg.binding
    sp, pl = pkg.world.mirror_box()
    cam = pkg.world.initial_camera()
    w, h, spp = 40, 24, 2
    start = initial_planes(ora, w, h)
    ctx.set_scene(sp, pl)
    ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if stream_form else B.FORM_AUTO)
    try:
        assert ctx.get_option(B.OPT_STREAM_STEP_CAP) == 1 << 16
        for cap in (1 << 16, 64, 3, 1):
            ctx.set_option(B.OPT_STREAM_STEP_CAP, cap)
            ctx.resize(w, h)
            ctx.upload_state(*start)
            ctx.reset_stats()
            ctx.render(cam, 15, spp, pkg.STREAMS)
            got, st = ctx.download_state(), ctx.stats()
            want, live, cut = ora.render_streams(sp, pl, cam, w, h, cap, spp, start, want_truncated=True)
            assert_planes_equal(got, want, "mirror box, cap %d" % cap)
            assert st["live_bounces"] == live and st["stream_rays_truncated"] == cut and st["stream_rays_dropped"] == 0
            if cap == 1 << 16:
                assert cut == 0 and 200 < st["stream_iterations"] < 5000
            else:
                assert cut > 0
    finally:
        ctx.set_option(B.OPT_STREAM_STEP_CAP, 1 << 16)
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)


@pytest.mark.parametrize("scene_name", ["s16", "glass"])
def test_sample_chunks_of_the_streams_kernels(pkg, ora, scene_name):
    """PTMI_OPT_SPP_CHUNKS on the per-pixel Streams kernels (chain and tree walk): chained copies of the tile grid, each
    a slice of the samples -- bit-identical to the oracle."""
    B = pkg.binding
    scene = pkg.world.scene16() if scene_name == "s16" else pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 136, 70, 11
    start = initial_planes(ora, w, h)
    with pkg.Context(0) as c:
        c.set_option(B.OPT_SPP_CHUNKS, 4)
        c.set_scene(*scene)
        c.resize(w, h)
        c.upload_state(*start)
        c.render(cam, 15, spp, pkg.STREAMS)
        got = c.download_state()
        live = c.stats()["live_bounces"]
    if scene_name == "s16":
        want, live_ref = ora.render_streams(scene[0], scene[1], cam, w, h, CAP, spp, start)
    else:
        want, live_ref = ora.render_streams_tree(scene[0], scene[1], cam, w, h, CAP, spp, start)[:2]
    assert_planes_equal(got, want, "streams in 4 sample chunks (%s)" % scene_name)
    assert live == live_ref


@pytest.mark.parametrize("stream_form", [False, True])
def test_stream_iterations_is_per_launch(ctx, pkg, ora, stream_form):
    """ptmi_stats.stream_iterations is the deepest traceStep of the LAST launch (sharded maxima, all shards cleared per launch):
    a deep launch (the mirror box: hundreds of steps) followed by a shallow one must report the shallow figure."""
    B = pkg.binding
    cam = pkg.world.initial_camera()
    w, h = 72, 40
    start = initial_planes(ora, w, h)
    ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if stream_form else B.FORM_AUTO)
    try:
        ctx.set_scene(*pkg.world.mirror_box())
        ctx.resize(w, h)
        ctx.upload_state(*start)
        ctx.render(cam, 15, 1, pkg.STREAMS)
        deep = ctx.stats()["stream_iterations"]
        ctx.set_scene(*pkg.world.main_scene())
        ctx.upload_state(*start)
        ctx.render(cam, 15, 1, pkg.STREAMS)
        shallow = ctx.stats()["stream_iterations"]
    finally:
        ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
    assert deep > 200 and 1 <= shallow < 64


@pytest.mark.parametrize("spp", [256, 32])
def test_stream_form_on_the_parts_of_a_striped_image(pkg, spp):
    """What one rank of a multi-GPU job renders through the stream form: a part of a row-striped image (8 parts, 10-row
    stripes) -- at 256 spp, with the pixels' sample chains cut into four ordered passes (PTMI_OPT_ORDERED_PASSES = 4: few, long items per
    lane; the caller's explicit choice since 0.6, never automatic), and at 32 spp, where a pixel's chain is one item and the second launch
    leaves the cheap end of the part's dispatch order to the per-pixel kernel.  Every part: all seven planes bit-identical to the per-pixel chain kernel on the same part, which
    the oracle pins; the parts together hold every row of the image once."""
    B = pkg.binding
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h, n_parts, stripe = 1280, 720, 8, 10
    rows_seen = 0
    for part in (0, 3, 7):
        with pkg.Context(0) as chain, pkg.Context(0) as stream:
            for c in (chain, stream):
                c.set_scene(sp, pl)
                c.set_partition(stripe, n_parts, part)
                c.resize(w, h)
                c.init_output(0xC0FFEE)
            stream.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            if spp >= 256:
                stream.set_option(B.OPT_ORDERED_PASSES, 4)
            assert chain.local_rows == stream.local_rows == h // n_parts
            for launch in range(3):
                chain.render(cam, 8, spp, pkg.STREAMS)
                stream.render(cam, 8, spp, pkg.STREAMS)
                assert_planes_equal(stream.download_state(), chain.download_state(), "part %d of %d, %d spp, launch %d" % (part, n_parts, spp, launch))
            assert stream.stats()["live_bounces"] == chain.stats()["live_bounces"]
            rows_seen += stream.local_rows
    assert rows_seen == 3 * (h // n_parts)


def test_the_stream_forms_per_pixel_tail_changes_no_bit(pkg):
    """Without GLASS the stream form leaves the cheap end of its dispatch order to the per-pixel chain kernel, launched beside the
    persistent kernel (its waves fill the slots that the persistent waves leave as they end).  Where the boundary lies --
    PTMI_OPT_STREAM_TAIL thousandths of the recorded cost: none, the default, nearly everything -- must change no bit: four launches
    each (the order and the boundary are rebuilt before the second and the third), whole pixel chains and ordered passes of 4
    samples, all seven planes against the chain kernel."""
    B = pkg.binding
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h, spp = 1280, 720, 16
    with pkg.Context(0) as chain:
        chain.set_scene(sp, pl)
        chain.resize(w, h)
        chain.init_output(0xABCDEF)
        want = []
        for _ in range(4):
            chain.render(cam, 8, spp, pkg.STREAMS)
            want.append(chain.download_state())
        live = chain.stats()["live_bounces"]
    for tail in (0, 150, 950):
        for batch in (0, 4):
            with pkg.Context(0) as c:
                c.set_scene(sp, pl)
                c.resize(w, h)
                c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
                c.set_option(B.OPT_STREAM_BATCH, batch)
                c.set_option(B.OPT_STREAM_TAIL, tail)
                assert c.get_option(B.OPT_STREAM_TAIL) == tail
                c.init_output(0xABCDEF)
                for k in range(4):
                    c.render(cam, 8, spp, pkg.STREAMS)
                    assert_planes_equal(c.download_state(), want[k], "tail %s, items of %s samples, launch %d" % (tail, batch or "all", k))
                assert c.stats()["live_bounces"] == live
