"""The stream ("wavefront") form of render Streams: ballot/prefix compaction of children, float atomics
for permute (+).  Without GLASS it must equal the per-pixel chain kernel and the oracle BIT FOR BIT (one
adder per colour word per launch).  With the build-defined GLASS extension (no reference semantics; spec in
oracle/pt_oracle.c glass_children) several rays of a pixel add in one launch in undefined order -- as in
Accelerate's permute -- so colours are compared within north_star's 1e-4 relative; seeds stay exact."""
import os

import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4
CAP = 1 << 16      # PTMI_OPT_STREAM_STEP_CAP default, both forms (the reference has no cap, Trace.hs:166-170)


def render(ctx, pkg, scene, cam, w, h, spp, start, variant=0, stream_form=False):
    B = pkg.binding
    ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM if stream_form else B.FORM_AUTO)
    ctx.set_variant(variant)
    ctx.set_scene(*scene)
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.reset_stats()
    ctx.render(cam, 15, spp, pkg.STREAMS)
    out, st = ctx.download_state(), ctx.stats()
    ctx.set_variant(0)
    ctx.set_option(B.OPT_STREAMS_FORM, B.FORM_AUTO)
    return out, st


@pytest.mark.parametrize("scene_name,w,h,spp", [("main", 96, 64, 2), ("s16", 120, 67, 3)])
def test_wavefront_without_glass_is_bit_exact(ctx, pkg, ora, scene_name, w, h, spp):
    scene = pkg.world.main_scene() if scene_name == "main" else pkg.world.scene16()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, w, h)
    got, st = render(ctx, pkg, scene, cam, w, h, spp, start, variant=9)          # force the stream form
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start)
    assert_planes_equal(got, want, "wavefront vs oracle stream")
    chain, _ = ora.render_streams(scene[0], scene[1], cam, w, h, 1 << 16, spp, start)
    assert_planes_equal(got, chain, "wavefront vs per-pixel chain")
    assert st["live_bounces"] == live and st["stream_iterations"] == steps and st["stream_rays_dropped"] == 0 == dropped


def test_glass_scene_tree_walk_is_bit_exact(ctx, pkg, ora):
    """The default for scenes with GLASS: the per-pixel tree walk.  One adder per colour word, additions in a defined
    order (depth first, reflection before refraction) -> equal to the oracle's tree order BIT FOR BIT, and within
    rounding of the stream order."""
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    for w, h, spp in ((128, 72, 3), (61, 33, 5)):
        start = initial_planes(ora, w, h)
        got, st = render(ctx, pkg, scene, cam, w, h, spp, start)
        want, live, dropped, longest, cut = ora.render_streams_tree(scene[0], scene[1], cam, w, h, CAP, spp, start)
        assert_planes_equal(got, want, "glass tree walk %dx%d" % (w, h))
        assert st["live_bounces"] == live and st["stream_rays_dropped"] == dropped == 0 and st["stream_rays_truncated"] == cut == 0
        assert st["stream_iterations"] == longest
        stream = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start, capacity_factor=8)[0]
        for a, b in zip(got[:3], stream[:3]):
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL


def test_glass_scene_stream_form_within_tolerance_and_seeds_exact(ctx, pkg, ora):
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 128, 72, 3
    start = initial_planes(ora, w, h)
    got, st = render(ctx, pkg, scene, cam, w, h, spp, start, stream_form=True)
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start)
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)                       # updateSeed: integer, exact
    assert st["live_bounces"] == live and st["stream_rays_dropped"] == dropped == 0
    assert st["stream_iterations"] == steps
    split = live - ora.render_streams_wavefront(pkg.world.scene16()[0], scene[1], cam, w, h, CAP, spp, start)[1]
    assert split != 0                                      # the scene really splits rays
    for a, b in zip(got[:3], want[:3]):
        scale = np.maximum(np.abs(b), 1e-3)
        assert np.max(np.abs(a - b) / scale) <= REL_TOL
    # and the glass changed the picture
    plain, _, _, _ = ora.render_streams_wavefront(pkg.world.scene16()[0], scene[1], cam, w, h, CAP, spp, start)
    assert not np.array_equal(plain[0], want[0])


def test_inline_refuses_glass(ctx, pkg):
    """"features that require diverging rays like light refraction" need the stream algorithm (Trace.hs:56-67)."""
    ctx.set_scene(*pkg.world.glass_scene())
    ctx.resize(16, 16)
    with pytest.raises(pkg.PtmiError) as e:
        ctx.render(pkg.world.initial_camera(), 8, 1, pkg.INLINE)
    assert e.value.code == -1
    ctx.render(pkg.world.initial_camera(), 8, 1, pkg.STREAMS)
    ctx.synchronize()


def test_stream_batch_option_trades_order_for_shorter_items(ctx, pkg, ora):
    """PTMI_OPT_STREAM_BATCH > 0 without GLASS: a pixel's samples are cut into items of that many samples which run in
    whatever lanes take them, so the order of a pixel's additions is no longer the sample order -- colours agree to
    rounding, the RNG planes stay exact, nothing is lost."""
    B = pkg.binding
    scene = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h, spp = 120, 67, 7
    start = initial_planes(ora, w, h)
    ctx.set_option(B.OPT_STREAM_BATCH, 4)
    ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_KEEP_ACCUMULATOR)      # (under `combine new old` a pixel's samples are one serial chain: the option is ignored)
    try:
        got, st = render(ctx, pkg, scene, cam, w, h, spp, start, stream_form=True)
    finally:
        ctx.set_option(B.OPT_STREAM_BATCH, 0)
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_AUTO)
    want, live = ora.render_streams(scene[0], scene[1], cam, w, h, CAP, spp, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)
    assert st["live_bounces"] == live and st["stream_rays_dropped"] == 0 and st["stream_rays_truncated"] == 0
    for a, b in zip(got[:3], want[:3]):
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL


def test_overflow_levels_with_tiny_rings(pkg, ora, tmp_path_factory):
    """The rare path of the stream form: a child that finds its wave's ring AND its wave's spill queue full travels through the
    overflow stream and is traced by a later launch (streams_level_kernel), level after level.  With the product's sizes (16 ring
    records, 4 096 spill records per wave) that practically never happens, so this test builds the library with a ring of 2 and a
    spill queue of 4 records: same rays, same counts, same seeds, colours within the tolerance of the undefined addition order."""
    B = pkg.binding
    out = os.path.join(str(tmp_path_factory.mktemp("tinyrings")), "libptmi_tinyrings.so")
    lib = B.open_library(pkg._build.build_lib(out=out, extra_flags=["-DPTMI_RING=2", "-DPTMI_SPILL=4"]))
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 160, 96, 4
    start = initial_planes(ora, w, h)
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start, capacity_factor=16)
    assert dropped == 0
    for batch in (0, 1):
        with pkg.Context(0, library=lib) as c:
            c.set_scene(*scene)
            c.resize(w, h)
            c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
            c.set_option(B.OPT_STREAM_CAPACITY, 16)
            c.set_option(B.OPT_STREAM_BATCH, batch)
            c.upload_state(*start)
            c.render(cam, 15, spp, pkg.STREAMS)
            got, st = c.download_state(), c.stats()
        for a, b in zip(got[3:], want[3:]):
            assert np.array_equal(a, b)
        assert st["live_bounces"] == live and st["stream_rays_dropped"] == 0 and st["stream_iterations"] == steps
        assert st["stream_rays_spilled"] > live // 50            # the tiny ring and spill queue really overflowed ...
        assert st["stream_rays_overflowed"] > live // 100        # ... and later launches traced what they could not hold
        for a, b in zip(got[:3], want[:3]):
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL


def test_a_call_that_would_drop_children_is_redone_with_longer_streams(pkg, ora, tmp_path_factory):
    """`expand` (Trace.hs:284-293) makes its vectors as long as the step needs.  The overflow streams have a capacity (PTMI_OPT_STREAM_CAPACITY
    rays per pixel); a call that would DROP children puts the colour planes back, doubles the streams and runs again, until nothing is dropped
    (or 64 rays per pixel / the device's memory are reached).  Forced here with a ring of 2 and a spill queue of 4 records per wave and ONE ray
    per pixel of overflow stream: the first attempt drops thousands of children; the call still returns the oracle's rays, counts and seeds,
    says stream_rays_dropped == 0, and the context keeps the capacity it grew to (visible through ptmi_get_option; a later call does not redo)."""
    B = pkg.binding
    out = os.path.join(str(tmp_path_factory.mktemp("tinyrings_grow")), "libptmi_tinyrings.so")
    lib = B.open_library(pkg._build.build_lib(out=out, extra_flags=["-DPTMI_RING=2", "-DPTMI_SPILL=4"]))
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 800, 600, 4                                  # 480 000 pixels: more than the streams' floor (the waves' static blocks, 393 472 slots), so one ray per pixel IS the capacity
    start = initial_planes(ora, w, h)
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start, capacity_factor=64)
    assert dropped == 0
    with pkg.Context(0, library=lib) as c:
        c.set_scene(*scene)
        c.resize(w, h)
        c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        c.set_option(B.OPT_STREAM_CAPACITY, 1)
        assert c.get_option(B.OPT_STREAM_CAPACITY) == 1
        c.upload_state(*start)
        c.render(cam, 15, spp, pkg.STREAMS)
        got, st = c.download_state(), c.stats()
        grown = c.get_option(B.OPT_STREAM_CAPACITY)
        assert grown > 1, "one ray per pixel held every child of a tiny-ring build: the test no longer forces the growth"
        for a, b in zip(got[3:], want[3:]):
            assert np.array_equal(a, b)
        assert st["live_bounces"] == live and st["stream_rays_dropped"] == 0 and st["stream_iterations"] == steps
        for a, b in zip(got[:3], want[:3]):
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL
        # the same samples again from the same state: the grown streams hold them at the first attempt, the colour is not added twice
        c.upload_state(*start)
        c.reset_stats()
        c.render(cam, 15, spp, pkg.STREAMS)
        again, st2 = c.download_state(), c.stats()
        assert c.get_option(B.OPT_STREAM_CAPACITY) == grown
        assert st2["live_bounces"] == live and st2["stream_rays_dropped"] == 0
        for a, b in zip(again[:3], want[:3]):
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL
        c.set_option(B.OPT_STREAM_CAPACITY, 2)                    # setting the option starts over from the value given
        assert c.get_option(B.OPT_STREAM_CAPACITY) == 2


@pytest.mark.parametrize("handoff", ["fenced", "fence_free"])
@pytest.mark.parametrize("rule", ["auto", "keep"])
def test_ordered_passes_inside_one_launch_are_bit_exact(ctx, pkg, ora, rule, handoff):
    """Without a ray-splitting material the stream form cuts a pixel's samples into ORDERED passes inside its one launch (a
    pixel's seven words travel through the planes from the lane that rendered one pass to whichever lane takes the next; a pass
    is handed out once the previous one has been published): here 64 samples as 4 items of 16, as 2 of 32 and as 32 of 2
    (PTMI_OPT_STREAM_BATCH under the result's-seed rule, where a pixel's samples are one serial chain) -- bit-identical to the oracle, both seed
    rules, image with and without whole tiles, through BOTH hand-offs (PTMI_OPT_PASS_HANDOFF): release / acquire once per region and pass
    (the default since 0.6; 32 passes of 2 samples put many chunks of a wave in flight at once), and the fence-free write-through form."""
    B = pkg.binding
    scene = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    spp = 64
    for w, h in ((200, 120), (61, 13)):
        start = initial_planes(ora, w, h)
        seed_rule = ora.SEED_KEEP_ACCUMULATOR if rule == "keep" else None
        want, live = ora.render_streams(scene[0], scene[1], cam, w, h, CAP, spp, start, seed_rule=seed_rule)
        for batch in (16, 32, 2):
            with pkg.Context(0) as c:
                c.set_scene(*scene)
                c.resize(w, h)
                c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
                c.set_option(B.OPT_PASS_HANDOFF, B.HANDOFF_FENCED if handoff == "fenced" else B.HANDOFF_FENCE_FREE)
                if rule == "keep":                               # (under the keep rule a batch selects the unordered split kernel instead)
                    c.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_KEEP_ACCUMULATOR)
                    c.set_option(B.OPT_ORDERED_PASSES, spp // batch)
                else:
                    c.set_option(B.OPT_STREAM_BATCH, batch)
                c.upload_state(*start)
                c.render(cam, 15, spp, pkg.STREAMS)
                got, st = c.download_state(), c.stats()
            assert_planes_equal(got, want, "ordered passes of %d samples, %dx%d, %s" % (batch, w, h, rule))
            assert st["live_bounces"] == live


def test_stream_form_with_a_scene_too_big_for_lds(ctx, pkg, ora):
    """A scene of 300 primitives (a third of them GLASS) is read through scalar loads instead of LDS in every kernel of the stream
    form (primary, split, overflow levels); same bar as for the small glass scene: counts and RNG planes exact, colours to 1e-4."""
    B = pkg.binding
    r = np.random.default_rng(300)
    wd = pkg.world
    n = 298
    spheres = np.zeros(n, wd.SPHERE_DTYPE)
    spheres["position"] = r.uniform(-25, 25, (n, 3)) + np.array([0, 3, -25])
    spheres["radius"] = r.uniform(0.3, 2.0, n)
    spheres["color"] = r.uniform(0.2, 1, (n, 3))
    spheres["illuminance"] = np.where(r.random(n) < 0.1, 20.0, 0.0)
    spheres["brdf_tag"] = np.where(r.random(n) < 0.33, wd.GLASS, wd.GLOSSY)
    spheres["brdf_param"] = np.where(spheres["brdf_tag"] == wd.GLASS, 1.5, r.uniform(0.2, 1.0, n))
    _, planes = wd.main_scene()
    planes = planes.copy()
    planes["brdf_tag"] = wd.GLOSSY                             # non-negative contributions only: a relative tolerance needs them
    planes["brdf_param"] = 0.7
    cam = wd.initial_camera()
    w, h, spp, cap = 144, 80, 3, 12
    start = initial_planes(ora, w, h)
    want, live, dropped, steps, cut = ora.render_streams_wavefront(spheres, planes, cam, w, h, cap, spp, start, capacity_factor=64, want_truncated=True)
    assert dropped == 0
    ctx.set_option(B.OPT_STREAM_STEP_CAP, cap)
    ctx.set_option(B.OPT_STREAM_CAPACITY, 64)
    try:
        got, st = render(ctx, pkg, (spheres, planes), cam, w, h, spp, start, stream_form=True)
    finally:
        ctx.set_option(B.OPT_STREAM_STEP_CAP, 1 << 16)
        ctx.set_option(B.OPT_STREAM_CAPACITY, 4)
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)
    assert st["live_bounces"] == live and st["stream_rays_dropped"] == 0 and st["stream_rays_truncated"] == cut
    for a, b in zip(got[:3], want[:3]):
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 + 1e-4 * np.max(np.abs(b)))) <= REL_TOL


@pytest.mark.parametrize("graded,glass_batch", [(1, 0), (0, 0), (1, 4), (1, 16), (0, 64)])
def test_the_split_kernels_scheduling_knobs_change_no_ray_and_no_seed(ctx, pkg, ora, graded, glass_batch):
    """PTMI_OPT_STREAM_GRADED (passes that shrink towards the end of the launch, the default, or round 3's uniform ones) and
    PTMI_OPT_GLASS_BATCH (GLASS hits parked in their lanes until that many are pending in the wave) decide WHEN a ray is traced and
    by which lane, never WHICH rays exist: counts and RNG planes equal the oracle's exactly, colours within the tolerance of the
    undefined order of a pixel's additions.  13 samples: graded passes of 3, 2, 2, 2, 2, 2 at this size."""
    B = pkg.binding
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 128, 72, 13
    start = initial_planes(ora, w, h)
    ctx.set_option(B.OPT_STREAM_GRADED, graded)
    ctx.set_option(B.OPT_GLASS_BATCH, glass_batch)
    try:
        assert ctx.get_option(B.OPT_STREAM_GRADED) == graded and ctx.get_option(B.OPT_GLASS_BATCH) == glass_batch
        got, st = render(ctx, pkg, scene, cam, w, h, spp, start, stream_form=True)
    finally:
        ctx.set_option(B.OPT_STREAM_GRADED, 1)
        ctx.set_option(B.OPT_GLASS_BATCH, 0)
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start)
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)
    assert st["live_bounces"] == live and st["stream_rays_dropped"] == dropped == 0 and st["stream_iterations"] == steps
    for a, b in zip(got[:3], want[:3]):
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL


@pytest.mark.parametrize("order", [1, 2, 3, 102, 104, 164])
def test_the_ticket_order_changes_no_ray_and_no_seed(ctx, pkg, ora, order):
    """PTMI_OPT_STREAM_PASS_GROUPS: pass by pass (1), the last two or three passes as one group, pairs, groups of four behind two passes
    in pass order, everything region by region (164) -- six graded passes of 3, 2, 2, 2, 2, 2 samples at this size.  Every item is handed out
    exactly once whatever the order: counts and RNG planes equal the oracle's, colours within the tolerance of the undefined order of additions."""
    B = pkg.binding
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 128, 72, 13
    start = initial_planes(ora, w, h)
    ctx.set_option(B.OPT_STREAM_PASS_GROUPS, order)
    try:
        assert ctx.get_option(B.OPT_STREAM_PASS_GROUPS) == order
        got, st = render(ctx, pkg, scene, cam, w, h, spp, start, stream_form=True)
    finally:
        ctx.set_option(B.OPT_STREAM_PASS_GROUPS, 0)
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start)
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)
    assert st["live_bounces"] == live and st["stream_rays_dropped"] == dropped == 0 and st["stream_iterations"] == steps
    for a, b in zip(got[:3], want[:3]):
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL


def test_a_small_snapshot_budget_merges_the_last_passes_and_changes_no_ray(ctx, pkg, ora):
    """PTMI_OPT_SNAPSHOT_BUDGET_MB: the seed snapshots are passes x record slots x 16 bytes (here 144 regions x 128 slots: 295 KB per pass, six
    graded passes of 3, 2, 2, 2, 2, 2 samples); a budget of 1 MB holds three, so the last four passes run as one -- same rays, same seeds, same
    counts as the oracle's.  A budget that does not hold ONE pass is PTMI_ELIMIT, and the context renders again once it is raised."""
    B = pkg.binding
    scene = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h, spp = 128, 72, 13
    start = initial_planes(ora, w, h)
    ctx.set_option(B.OPT_SNAPSHOT_BUDGET_MB, 1)
    try:
        assert ctx.get_option(B.OPT_SNAPSHOT_BUDGET_MB) == 1
        got, st = render(ctx, pkg, scene, cam, w, h, spp, start, stream_form=True)
        with pytest.raises(pkg.PtmiError) as e:
            render(ctx, pkg, scene, cam, 640, 360, 2, initial_planes(ora, 640, 360), stream_form=True)     # 7.4 MB per pass
        assert e.value.code == B.PTMI_ELIMIT
    finally:
        ctx.set_option(B.OPT_SNAPSHOT_BUDGET_MB, 0)
    want, live, dropped, steps = ora.render_streams_wavefront(scene[0], scene[1], cam, w, h, CAP, spp, start)
    for a, b in zip(got[3:], want[3:]):
        assert np.array_equal(a, b)
    assert st["live_bounces"] == live and st["stream_rays_dropped"] == dropped == 0 and st["stream_iterations"] == steps
    for a, b in zip(got[:3], want[:3]):
        assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= REL_TOL
    again, _ = render(ctx, pkg, scene, cam, w, h, spp, start, stream_form=True)          # the default budget: six passes
    for a, b in zip(again[3:], want[3:]):
        assert np.array_equal(a, b)


def test_options_of_the_stream_form_are_range_checked_and_visible(pkg):
    B = pkg.binding
    with pkg.Context(0) as c:
        defaults = {B.OPT_STREAM_TAIL: -1, B.OPT_ORDERED_PASSES: 0, B.OPT_GLASS_BATCH: 0, B.OPT_STREAM_GRADED: 1, B.OPT_SNAPSHOT_BUDGET_MB: 0,
                    B.OPT_STREAM_PASS_GROUPS: 0, B.OPT_PASS_HANDOFF: B.HANDOFF_FENCED, B.OPT_CHAIN_SLOTS: 0}
        for opt, value in defaults.items():
            assert c.get_option(opt) == value
        for opt, bad in ((B.OPT_STREAM_TAIL, -2), (B.OPT_STREAM_TAIL, 1001), (B.OPT_ORDERED_PASSES, -1), (B.OPT_ORDERED_PASSES, 65),
                         (B.OPT_GLASS_BATCH, 65), (B.OPT_STREAM_GRADED, 2), (B.OPT_SNAPSHOT_BUDGET_MB, -1), (B.OPT_SNAPSHOT_BUDGET_MB, (1 << 20) + 1),
                         (B.OPT_STREAM_PASS_GROUPS, -1), (B.OPT_STREAM_PASS_GROUPS, 65), (B.OPT_STREAM_PASS_GROUPS, 101), (B.OPT_STREAM_PASS_GROUPS, 165),
                         (B.OPT_PASS_HANDOFF, 2), (B.OPT_PASS_HANDOFF, -1), (B.OPT_CHAIN_SLOTS, 1), (B.OPT_CHAIN_SLOTS, 4097)):
            with pytest.raises(pkg.PtmiError) as e:
                c.set_option(opt, bad)
            assert e.value.code == B.PTMI_EINVAL
            assert c.get_option(opt) == defaults[opt]
