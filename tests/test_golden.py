"""Committed fixtures (tests/golden/*.npz, provenance: self-oracle, see make_golden.py):
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
import glob
import os

import numpy as np
import pytest

from conftest import assert_planes_equal

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    start = [z["in_%d" % i] for i in range(7)]
    want = [z["out_%d" % i] for i in range(7)]
    return z, start, want


def test_fixtures_exist():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(ora, path):
    z, start, want = load(path)
    w, h, limit, spp = int(z["width"]), int(z["height"]), int(z["limit"]), int(z["spp"])
    if str(z["algorithm"]) == "inline":
        got, live = ora.render_inline(z["spheres"], z["planes"], z["camera"], w, h, limit, spp, start)
    elif str(z["algorithm"]) == "streams":           # the library's default seed rule
        got, live = ora.render_streams(z["spheres"], z["planes"], z["camera"], w, h, limit, spp, start)
    elif str(z["algorithm"]) == "streams_keep":
        got, live = ora.render_streams(z["spheres"], z["planes"], z["camera"], w, h, limit, spp, start, seed_rule=ora.SEED_KEEP_ACCUMULATOR)
    else:
        got, live, _dropped, _steps = ora.render_streams_wavefront(z["spheres"], z["planes"], z["camera"], w, h, limit, spp, start)
    assert_planes_equal(got, want, os.path.basename(path))
    assert live == int(z["live"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_gpu_reproduces_golden(ctx, pkg, path):
    z, start, want = load(path)
    w, h, limit, spp = int(z["width"]), int(z["height"]), int(z["limit"]), int(z["spp"])
    alg = pkg.INLINE if str(z["algorithm"]) == "inline" else pkg.STREAMS
    ctx.set_scene(z["spheres"], z["planes"])
    ctx.resize(w, h)
    ctx.upload_state(*start)
    ctx.reset_stats()
    B = pkg.binding
    ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_KEEP_ACCUMULATOR if str(z["algorithm"]) == "streams_keep" else B.SEED_AUTO)
    try:
        ctx.render(z["camera"], limit, spp, alg)
        got = ctx.download_state()
    finally:
        ctx.set_option(B.OPT_STREAMS_SEED_RULE, B.SEED_AUTO)
    if str(z["algorithm"]) == "wavefront":
        # GLASS: several rays of a pixel add in one launch, in undefined order (as in Accelerate's permute)
        assert_planes_equal(got[3:], want[3:] , os.path.basename(path))
        for a, b in zip(got[:3], want[:3]):
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= 1e-4
    else:
        assert_planes_equal(got, want, os.path.basename(path))
    assert ctx.stats()["live_bounces"] == int(z["live"])
