"""The chained closure on the device (include/ptmi.h, "the closure, chained": ptmi_render1_chained / ptmi_chain_*): compileFor's pure
`Camera -> (Int, RenderResult) -> (Int, RenderResult)` (app/Main.hs:83-84, :188-191) with runN's device residency -- a RenderResult is a
token, nothing crosses PCIe until somebody reads a plane -- against the oracle, bit for bit; and the properties that make a token a VALUE:
the input of a call stands, an old token still reads as it did, a state that had to leave the device is served from the host."""
import numpy as np
import pytest

from conftest import assert_planes_equal, initial_planes

pytestmark = pytest.mark.gpu
CAP = 1 << 16


@pytest.fixture()
def fresh(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def moved(pkg, cam, dx=0.75, droll=0.1):
    pos, rot = np.array(cam["position"], copy=True), np.array(cam["rotation"], copy=True)
    pos[0] += dx
    rot[0] += droll
    return pkg.world.camera(tuple(pos), tuple(rot), int(cam["fov"]))


def test_chain_of_100_calls_a_camera_move_and_a_reseed_equal_the_oracle(fresh, pkg, ora):
    """main's flow through the chained closure: seeds <- initialOutput; value = compute camera (0, seeds); 100 times `compute camera
    value` one sample each (computationLoop below its batching threshold, app/Main.hs:209-211); a reseed (:231); a camera move -- a fresh
    initialOutput and the moved camera (:306-319).  No plane crosses PCIe until the fetches; all seven planes equal the oracle's."""
    ctx = fresh
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h, limit = 96, 64, 15
    ctx.set_scene(sp, pl)
    tok = ctx.chain_init_output(w, h, 0x5EED1234)
    start = initial_planes(ora, w, h)
    assert_planes_equal(ctx.chain_fetch(tok, w, h), start, "chain_init_output")
    tokens = [tok]
    for _ in range(100):
        tok, fetched = ctx.render1_chained(cam, limit, w, h, tok)
        assert fetched == {}
        tokens.append(tok)
    assert len(set(tokens)) == 101 and 0 not in tokens
    want, _ = ora.render_inline(sp, pl, cam, w, h, limit, 100, start)
    assert_planes_equal(ctx.chain_fetch(tok, w, h), want, "100 chained calls")
    info = ctx.chain_info()
    assert info["renders_chained"] == 100 and info["renders_uploaded"] == 0 and info["renders_in_place"] == 0
    # the value after 37 calls is still the value after 37 calls (it may have left the device meanwhile)
    want37, _ = ora.render_inline(sp, pl, cam, w, h, limit, 37, start)
    assert_planes_equal(ctx.chain_fetch(tokens[37], w, h), want37, "the 37th value, read after the 100th was made")
    # reseed: colour kept, every RNG state replaced (Util.hs:134-135)
    tok_r = ctx.chain_reseed(4242, w, h, tok)
    reseeded = list(want[:3]) + [s.reshape(h, w) for s in ora.gen_seeds(4242, 0, w * h)]
    assert_planes_equal(ctx.chain_fetch(tok_r, w, h), reseeded, "chain_reseed")
    assert_planes_equal(ctx.chain_fetch(tok, w, h), want, "reseed left its input as it was")
    tok_r2, _ = ctx.render1_chained(cam, limit, w, h, tok_r)
    want_r2, _ = ora.render_inline(sp, pl, cam, w, h, limit, 1, reseeded)
    assert_planes_equal(ctx.chain_fetch(tok_r2, w, h), want_r2, "a sample after the reseed")
    # camera move: emptyOutput <- initialOutput; compute updatedCamera (0, emptyOutput)
    cam2 = moved(pkg, cam)
    tok_m = ctx.chain_init_output(w, h, 99)
    start2 = initial_planes(ora, w, h, 99)
    for _ in range(3):
        tok_m, _ = ctx.render1_chained(cam2, limit, w, h, tok_m)
    want_m, _ = ora.render_inline(sp, pl, cam2, w, h, limit, 3, start2)
    assert_planes_equal(ctx.chain_fetch(tok_m, w, h), want_m, "three samples after the camera move")
    for t in tokens:
        ctx.chain_release(t)
    left = ctx.chain_info()
    assert left["states_on_device"] + left["states_on_host"] == 6      # the reseeded value and its successor; initialOutput after the move and its three successors


def test_the_closure_is_a_function_of_its_token(fresh, pkg, ora):
    """Two calls from ONE token give the same planes twice, and equal one call of ptmi_render1 on the host planes; the planes asked for
    with the call (planes_out) equal a later fetch."""
    ctx = fresh
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h = 70, 41
    ctx.set_scene(sp, pl)
    t0 = ctx.chain_init_output(w, h, 5)
    t1, _ = ctx.render1_chained(cam, 8, w, h, t0)
    ta, got_a = ctx.render1_chained(cam, 8, w, h, t1, fetch=("r", "g", "b"))
    tb, _ = ctx.render1_chained(cam, 8, w, h, t1)
    assert len({t0, t1, ta, tb}) == 4
    a, b = ctx.chain_fetch(ta, w, h), ctx.chain_fetch(tb, w, h)
    assert_planes_equal(a, b, "two calls from one token")
    assert_planes_equal([got_a["r"], got_a["g"], got_a["b"]], a[:3], "planes_out of the call")
    compat = ctx.render1(cam, 8, w, h, ctx.chain_fetch(t1, w, h))
    assert_planes_equal(a, compat, "chained call vs ptmi_render1")
    colour_only = ctx.chain_fetch(ta, w, h, "r b")
    assert len(colour_only) == 2 and np.array_equal(colour_only[1], a[2])


def test_an_input_that_is_not_a_held_state_takes_the_copy_path_and_is_exact(fresh, pkg, ora):
    """token 0, a released token, another context's token: the host planes given with the call are uploaded (ptmi_render1's path) and the
    result is the oracle's; without host planes: PTMI_ESTALE."""
    ctx = fresh
    B = pkg.binding
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 64, 48
    ctx.set_scene(sp, pl)
    start = initial_planes(ora, w, h, 31)
    start[0][:] = 0.5                                       # a RenderResult that did not come from this library
    want, _ = ora.render_inline(sp, pl, cam, w, h, 15, 1, start)
    t, _ = ctx.render1_chained(cam, 15, w, h, 0, planes_in=start)
    assert_planes_equal(ctx.chain_fetch(t, w, h), want, "token 0 + host planes")
    assert ctx.chain_info()["renders_uploaded"] == 1 and ctx.chain_info()["renders_chained"] == 0
    ctx.chain_release(t)
    ctx.chain_release(t)                                    # idempotent
    with pytest.raises(pkg.PtmiError) as e:
        ctx.chain_fetch(t, w, h)
    assert e.value.code == B.PTMI_ESTALE
    with pytest.raises(pkg.PtmiError) as e:
        ctx.render1_chained(cam, 15, w, h, t)                # released, and nothing to take its place
    assert e.value.code == B.PTMI_ESTALE
    t2, _ = ctx.render1_chained(cam, 15, w, h, t, planes_in=start)     # released, but the caller still has the planes
    assert_planes_equal(ctx.chain_fetch(t2, w, h), want, "stale token + host planes")
    with pkg.Context(0) as other:
        other.set_scene(sp, pl)
        foreign = other.chain_init_output(w, h, 1)
        with pytest.raises(pkg.PtmiError) as e:
            ctx.chain_fetch(foreign, w, h)
        assert e.value.code == B.PTMI_ESTALE
        assert foreign != ctx.chain_init_output(w, h, 1)
    with pytest.raises(pkg.PtmiError) as e:
        ctx.render1_chained(cam, 15, w + 1, h, t2)           # a state of another size under that token
    assert e.value.code == B.PTMI_EINVAL
    with pytest.raises(pkg.PtmiError) as e:
        ctx.render1_chained(cam, 15, w, h, 0)
    assert e.value.code == B.PTMI_ESTALE
    # reseed's copy path: colour planes from the host
    tr = ctx.chain_reseed(8, w, h, 0, colour_in=want[:3])
    assert_planes_equal(ctx.chain_fetch(tr, w, h), list(want[:3]) + [s.reshape(h, w) for s in ora.gen_seeds(8, 0, w * h)], "reseed from host colour")


def test_states_that_leave_the_device_are_served_from_the_host(fresh, pkg, ora):
    """PTMI_OPT_CHAIN_SLOTS = 2: a chain of 6 unreleased values keeps two on the device; the others moved to host memory and read back
    exactly; a call FROM an evicted value works (it is uploaded again) and does not disturb it."""
    ctx = fresh
    B = pkg.binding
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 50, 30
    ctx.set_scene(sp, pl)
    ctx.set_option(B.OPT_CHAIN_SLOTS, 2)
    start = initial_planes(ora, w, h, 77)
    toks = [ctx.chain_init_output(w, h, 77)]
    for _ in range(5):
        toks.append(ctx.render1_chained(cam, 15, w, h, toks[-1])[0])
    info = ctx.chain_info()
    assert info["states_on_device"] == 2 and info["states_on_host"] == 4 and info["evictions"] == 4 and info["device_slots"] == 2
    for k in (0, 2, 5):
        want, _ = ora.render_inline(sp, pl, cam, w, h, 15, k, start)
        assert_planes_equal(ctx.chain_fetch(toks[k], w, h), want, "value %d" % k)
    t, _ = ctx.render1_chained(cam, 15, w, h, toks[1])       # from a value that lives on the host
    want2, _ = ora.render_inline(sp, pl, cam, w, h, 15, 2, start)
    assert_planes_equal(ctx.chain_fetch(t, w, h), want2, "a call from an evicted value")
    want1, _ = ora.render_inline(sp, pl, cam, w, h, 15, 1, start)
    assert_planes_equal(ctx.chain_fetch(toks[1], w, h), want1, "... which stands")
    with pytest.raises(pkg.PtmiError):
        ctx.set_option(B.OPT_CHAIN_SLOTS, 1)


def test_consume_renders_in_place(fresh, pkg, ora):
    """PTMI_CHAIN_CONSUME: the caller gives the input up; no copy is made, the input token is gone, the result is the same."""
    ctx = fresh
    B = pkg.binding
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h = 81, 37
    ctx.set_scene(sp, pl)
    start = initial_planes(ora, w, h, 3)
    t = ctx.chain_init_output(w, h, 3)
    first = t
    for _ in range(4):
        t, _ = ctx.render1_chained(cam, 8, w, h, t, consume=True)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 8, 4, start)
    assert_planes_equal(ctx.chain_fetch(t, w, h), want, "four consuming calls")
    info = ctx.chain_info()
    assert info["renders_in_place"] == 4 and info["states_on_device"] == 1 and info["states_on_host"] == 0
    with pytest.raises(pkg.PtmiError) as e:
        ctx.chain_fetch(first, w, h)
    assert e.value.code == B.PTMI_ESTALE
    t2 = ctx.chain_reseed(12, w, h, t, consume=True)
    assert ctx.chain_info()["states_on_device"] == 1
    assert_planes_equal(ctx.chain_fetch(t2, w, h)[:3], want[:3], "consuming reseed keeps the colour")


def test_render_streams_through_the_chain(fresh, pkg, ora):
    """--variant streams (app/Main.hs:112-132) through the same closure."""
    ctx = fresh
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 72, 40
    ctx.set_scene(sp, pl)
    start = initial_planes(ora, w, h)
    t = ctx.chain_init_output(w, h, 0x5EED1234)
    for _ in range(3):
        t, _ = ctx.render1_chained(cam, 15, w, h, t, algorithm=pkg.STREAMS)
    want, _ = ora.render_streams(sp, pl, cam, w, h, CAP, 3, start)
    assert_planes_equal(ctx.chain_fetch(t, w, h), want, "three Streams samples, chained")


def test_the_resident_planes_are_untouched_by_the_chain(fresh, pkg, ora):
    ctx = fresh
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 40, 24
    ctx.set_scene(sp, pl)
    ctx.resize(w, h)
    ctx.init_output(1)
    ctx.render(cam, 15, 2)
    before = ctx.download_state()
    t = ctx.chain_init_output(w, h, 2)
    t, _ = ctx.render1_chained(cam, 15, w, h, t)
    assert_planes_equal(ctx.download_state(), before, "resident state")
    ctx.render(cam, 15, 1)
    want, _ = ora.render_inline(sp, pl, cam, w, h, 15, 3, initial_planes(ora, w, h, 1))
    assert_planes_equal(ctx.download_state(), want, "resident state, one more sample")


def test_random_walks_over_the_chain_api_equal_a_model_of_values(fresh, pkg, ora):
    """A token is a value: whatever sequence of calls -- render from ANY held token (not only the newest), reseed, consume, release, fetch, with
    three device slots so that states keep moving to the host and back -- every held token reads as the planes a pure model computes for it
    with the oracle (one sample = ora.render_inline on the parent's planes)."""
    ctx = fresh
    B = pkg.binding
    sp, pl = pkg.world.main_scene()
    cams = [pkg.world.initial_camera(), moved(pkg, pkg.world.initial_camera(), 0.4, -0.07)]
    w, h, limit = 24, 10, 6
    ctx.set_scene(sp, pl)
    ctx.set_option(B.OPT_CHAIN_SLOTS, 3)
    rng = np.random.default_rng(20261005)
    model = {}                                               # token -> seven planes

    def check(tok):
        assert_planes_equal(ctx.chain_fetch(tok, w, h), model[tok], "token %x" % tok)

    t = ctx.chain_init_output(w, h, 11)
    model[t] = initial_planes(ora, w, h, 11)
    ops = {"render": 0, "consume": 0, "reseed": 0, "release": 0, "fetch": 0, "init": 0, "from_host": 0}
    for step in range(400):
        held = sorted(model)
        r = rng.random()
        src = int(held[rng.integers(len(held))])
        cam = cams[int(rng.integers(2))]
        if r < 0.45:                                         # the closure, from any held value
            consume = rng.random() < 0.25
            new, got = ctx.render1_chained(cam, limit, w, h, src, consume=bool(consume), fetch=("g",) if rng.random() < 0.2 else ())
            want, _ = ora.render_inline(sp, pl, cam, w, h, limit, 1, model[src])
            model[new] = list(want)
            if "g" in got:
                assert np.array_equal(got["g"].view(np.uint32), want[1].view(np.uint32))
            if consume:
                del model[src]
                ops["consume"] += 1
            ops["render"] += 1
        elif r < 0.55:
            seed0 = int(rng.integers(1, 1 << 40))
            new = ctx.chain_reseed(seed0, w, h, src)
            model[new] = list(model[src][:3]) + [s.reshape(h, w) for s in ora.gen_seeds(seed0, 0, w * h)]
            ops["reseed"] += 1
        elif r < 0.62:                                       # a RenderResult from elsewhere: host planes, token 0
            planes = [p.copy() for p in model[src]]
            new, _ = ctx.render1_chained(cam, limit, w, h, 0, planes_in=planes)
            want, _ = ora.render_inline(sp, pl, cam, w, h, limit, 1, planes)
            model[new] = list(want)
            ops["from_host"] += 1
        elif r < 0.70:
            seed0 = int(rng.integers(1, 1 << 40))
            new = ctx.chain_init_output(w, h, seed0)
            model[new] = initial_planes(ora, w, h, seed0)
            ops["init"] += 1
        elif r < 0.85 and len(held) > 2:
            ctx.chain_release(src)
            del model[src]
            with pytest.raises(pkg.PtmiError):
                ctx.chain_fetch(src, w, h)
            ops["release"] += 1
        else:
            check(src)
            ops["fetch"] += 1
    for tok in model:
        check(tok)
    info = ctx.chain_info()
    assert info["states_on_device"] + info["states_on_host"] == len(model) and info["states_on_device"] <= 3
    assert info["evictions"] > 20 and ops["consume"] > 10 and ops["release"] > 10 and ops["from_host"] > 5, (info, ops)


def test_three_threads_share_the_closure_like_the_application(fresh, pkg, ora):
    """app/Main.hs:178-180: a computation thread applies the closure in a loop, the graphics thread reads the colour planes of whatever value
    the MVar holds, a third thread lets old values go -- all on ONE context, entered from three OS threads at once (ctypes drops the GIL
    inside the calls).  Every colour snapshot the reader took must be the oracle's planes after exactly that many samples."""
    import threading
    import time
    ctx = fresh
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h, limit, total = 64, 40, 15, 240
    ctx.set_scene(sp, pl)
    start = initial_planes(ora, w, h, 5)
    lock = threading.Lock()
    state = {"value": (0, ctx.chain_init_output(w, h, 5)), "old": [], "done": False, "error": None}
    snapshots = []

    def compute():
        try:
            for _ in range(total):
                k, tok = state["value"]
                new, _ = ctx.render1_chained(cam, limit, w, h, tok)
                with lock:
                    state["value"] = (k + 1, new)
                    state["old"].append(tok)
        except Exception as e:                               # noqa: BLE001
            state["error"] = e
        finally:
            state["done"] = True

    def graphics():
        try:
            while not state["done"]:
                with lock:
                    k, tok = state["value"]
                    # (the value in hand cannot be released under the reader: the releaser only takes what left `value` before)
                    r, g, b = ctx.chain_fetch(tok, w, h, "r g b")
                if not snapshots or snapshots[-1][0] != k:
                    snapshots.append((k, r, g, b))
                time.sleep(0.001)                            # (graphicsLoop does not spin either; a spinning reader starves the lock)
        except Exception as e:                               # noqa: BLE001
            state["error"] = e

    def releaser():
        try:
            while True:
                finished = state["done"]
                with lock:                                   # (all but the newest: nobody holds those any more; after the end, all of them)
                    olds, state["old"] = (state["old"], []) if finished else (state["old"][:-1], state["old"][-1:])
                for t in olds:
                    ctx.chain_release(t)
                if finished:
                    break
                time.sleep(0.001)
        except Exception as e:                               # noqa: BLE001
            state["error"] = e

    threads = [threading.Thread(target=f, daemon=True) for f in (compute, graphics, releaser)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a thread did not finish"
    assert state["error"] is None, state["error"]
    assert state["value"][0] == total and len(snapshots) >= 3
    picks = snapshots[:: max(1, len(snapshots) // 6)][:6] + [snapshots[-1]]
    for k, r, g, b in picks:
        want, _ = ora.render_inline(sp, pl, cam, w, h, limit, k, start)
        assert_planes_equal([r, g, b], want[:3], "the reader's snapshot after %d samples" % k)
    want, _ = ora.render_inline(sp, pl, cam, w, h, limit, total, start)
    assert_planes_equal(ctx.chain_fetch(state["value"][1], w, h), want, "the final value")
    assert ctx.chain_info()["renders_uploaded"] == 0
