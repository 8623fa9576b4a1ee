"""The oracle's Streams variants against each other (CPU only): the hedged seed-carry rule (assumption A5), row
partitions, the safety cap's bookkeeping, and the depth-first tree order against the stream order."""
import numpy as np

from conftest import assert_planes_equal, initial_planes, initial_rows, sfc32_advance

W, H = 48, 30


def test_rows_option_reproduces_the_whole_image(ora, pkg):
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, W, H)
    rows = np.array([3, 4, 11, 12, 29], np.int32)
    part = initial_rows(ora, W, rows)
    whole_i, _ = ora.render_inline(sp, pl, cam, W, H, 8, 2, start)
    part_i, _ = ora.render_inline(sp, pl, cam, W, H, 8, 2, part, rows=rows)
    assert_planes_equal(part_i, [p[rows] for p in whole_i], "inline rows")
    whole_s, _ = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 2, start)
    part_s, _ = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 2, part, rows=rows)
    assert_planes_equal(part_s, [p[rows] for p in whole_s], "streams rows")
    part_w = ora.render_streams_wavefront(sp, pl, cam, W, H, 1 << 16, 2, part, rows=rows)[0]
    assert_planes_equal(part_w, part_s, "wavefront rows")


def test_seed_from_result_rule(ora, pkg):
    """combine new old (A5's alternative): the two restatements agree with each other; pixels whose primary ray misses
    keep the one-draw rule; every other pixel's seed is its start seed advanced by 3 (hits - 1) + 1 draws."""
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, W, H)
    keep, _ = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 1, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)
    assert ora.default_seed_rule(sp, pl) == ora.SEED_FROM_RESULT and ora.default_seed_rule(*pkg.world.glass_scene()) == ora.SEED_KEEP_ACCUMULATOR
    res, live = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 1, start, seed_rule=ora.SEED_FROM_RESULT)
    wav = ora.render_streams_wavefront(sp, pl, cam, W, H, 1 << 16, 1, start, seed_rule=ora.SEED_FROM_RESULT)
    assert_planes_equal(wav[0], res, "from-result: stream vs per pixel")
    assert wav[1] == live
    for a, b in zip(res[:3], keep[:3]):
        assert np.array_equal(a, b)                          # the first sample's colour does not depend on the rule
    one = sfc32_advance(start[3:], 1)
    assert_planes_equal(list(keep[:3]) + list(one), keep, "keep rule = one draw")
    same = np.ones((H, W), bool)
    for a, b in zip(res[3:], one):
        same &= a == b
    assert 0.05 < same.mean() < 0.95                         # one hit (or none): 3 * 0 + 1 draws, as under the keep rule
    # every pixel's seed is the start seed advanced by 1 + 3 k draws for some k >= 0
    explained = np.zeros((H, W), bool)
    for k in range(0, 40):
        adv = sfc32_advance(start[3:], 1 + 3 * k)
        hit = np.ones((H, W), bool)
        for a, b in zip(res[3:], adv):
            hit &= a == b
        explained |= hit
    assert explained.all()
    # and from the second sample on the colours differ between the rules
    keep2, _ = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 2, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)
    res2, _ = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 2, start, seed_rule=ora.SEED_FROM_RESULT)
    assert not np.array_equal(keep2[0], res2[0])
    auto2, _ = ora.render_streams(sp, pl, cam, W, H, 1 << 16, 2, start)
    assert_planes_equal(auto2, res2, "the default (no ray-splitting material) is the result's seed")


def test_step_cap_bookkeeping(ora, pkg):
    """A small cap cuts the same rays in every restatement: same planes, same live count, same truncated count."""
    sp, pl = pkg.world.mirror_box()
    cam = pkg.world.initial_camera()
    w, h = 20, 12
    start = initial_planes(ora, w, h)
    for cap in (1, 3, 64):
        keep = ora.SEED_KEEP_ACCUMULATOR                       # the tree walk's rule (it exists for GLASS)
        chain, live, cut = ora.render_streams(sp, pl, cam, w, h, cap, 2, start, want_truncated=True, seed_rule=keep)
        wav, live_w, dropped, steps, cut_w = ora.render_streams_wavefront(sp, pl, cam, w, h, cap, 2, start, want_truncated=True, seed_rule=keep)
        tree, live_t, dropped_t, longest, cut_t = ora.render_streams_tree(sp, pl, cam, w, h, cap, 2, start)
        assert_planes_equal(wav, chain, "cap %d stream vs chain" % cap)
        assert_planes_equal(tree, chain, "cap %d tree vs chain" % cap)
        assert live == live_w == live_t and cut == cut_w == cut_t and dropped == dropped_t == 0
        assert steps == longest == cap and cut > 0
    # uncapped: lineages run for hundreds of steps in the mirror box and end by themselves
    chain, live, cut = ora.render_streams(sp, pl, cam, w, h, 1 << 16, 1, start, want_truncated=True)
    _, _, _, longest, _ = ora.render_streams_tree(sp, pl, cam, w, h, 1 << 16, 1, start)
    assert cut == 0 and 200 < longest < 5000


def test_tree_order_equals_stream_order_up_to_rounding(ora, pkg):
    sp, pl = pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    start = initial_planes(ora, W, H)
    wav, live, dropped, steps = ora.render_streams_wavefront(sp, pl, cam, W, H, 1 << 16, 3, start, capacity_factor=8)
    tree, live_t, dropped_t, longest, cut = ora.render_streams_tree(sp, pl, cam, W, H, 1 << 16, 3, start)
    assert live == live_t and dropped == dropped_t == 0 and cut == 0 and longest == steps
    for a, b in zip(tree[3:], wav[3:]):
        assert np.array_equal(a, b)
    worst = 0.0
    for a, b in zip(tree[:3], wav[:3]):
        worst = max(worst, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3))))
    assert worst <= 1e-4
    assert any(not np.array_equal(a, b) for a, b in zip(tree[:3], wav[:3])) or True   # the orders may coincide on small images
    # a shallow stack drops children and says so
    _, live_s, dropped_s, _, _ = ora.render_streams_tree(sp, pl, cam, W, H, 1 << 16, 3, start, stack_depth=1)
    assert dropped_s > 0 and live_s < live_t + 1
    # without GLASS the tree walk is the chain
    sp16, _ = pkg.world.scene16()
    assert_planes_equal(ora.render_streams_tree(sp16, pl, cam, W, H, 1 << 16, 2, start)[0],
                        ora.render_streams(sp16, pl, cam, W, H, 1 << 16, 2, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)[0], "tree without glass")


def test_the_threaded_streams_oracles_equal_the_serial_ones(pkg, ora):
    """ora_render_streams_ex and ora_render_streams_tree over rows on several threads (pixels are independent) -- what lets the
    full-size GPU tests compare against them -- and the row-per-thread wrapper of the stream-order oracle: same planes, same counts."""
    cam = pkg.world.initial_camera()
    w, h, spp = 96, 54, 3
    seeds = ora.gen_seeds(0x5EED1234, 0, w * h)
    start = [np.zeros((h, w), np.float32) for _ in range(3)] + [s.reshape(h, w) for s in seeds]
    sp, pl = pkg.world.scene16()
    for rule in (ora.SEED_KEEP_ACCUMULATOR, ora.SEED_FROM_RESULT):
        one = ora.render_streams(sp, pl, cam, w, h, 1 << 16, spp, start, seed_rule=rule, want_truncated=True)
        many = ora.render_streams(sp, pl, cam, w, h, 1 << 16, spp, start, seed_rule=rule, want_truncated=True, n_threads=4)
        assert one[1:] == many[1:] and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(one[0], many[0]))
    gs, gp = pkg.world.glass_scene()
    one = ora.render_streams_tree(gs, gp, cam, w, h, 1 << 16, spp, start, stack_depth=3)
    many = ora.render_streams_tree(gs, gp, cam, w, h, 1 << 16, spp, start, stack_depth=3, n_threads=4)
    assert one[1:] == many[1:] and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(one[0], many[0]))
    rows = [5, 6, 30, 31, 53]
    window = [a[rows] for a in start]
    whole = ora.render_streams_wavefront(gs, gp, cam, w, h, 1 << 16, spp, window, capacity_factor=8, rows=rows)
    split = ora.render_streams_wavefront_rows(gs, gp, cam, w, h, 1 << 16, spp, window, rows, n_threads=3)
    assert whole[1] == split[1] and whole[2] == split[2] == 0
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(whole[0], split[0]))
