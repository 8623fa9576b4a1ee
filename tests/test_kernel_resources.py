"""Register, scratch and LDS budgets of the render kernels (compiled here, no GPU needed).  The occupancy every
measurement in DESIGN.md rests on is decided by these numbers, and a few bytes of scratch inside a render loop have cost
tens of percent more than once: a kernel edit that breaks a budget should fail here, not in a profile."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def resources(tmp_path_factory):
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.collect(out_dir=str(tmp_path_factory.mktemp("isa")))


# kernel -> (VGPRs for its waves per SIMD, scratch bytes, static LDS bytes)
BUDGETS = {
    "render_inline_kernel<true, 0, 8>": (72, 16, 4096),          # 7 waves/SIMD; three registers spilled around the loop
    "render_inline_kernel<false, 0, 8>": (72, 32, 4096),         # scene through scalar loads (big scenes): one more pair around the loop
    "render_streams_kernel<true, 8>": (72, 16, 3328),            # 7 waves/SIMD
    "render_streams_kernel<false, 8>": (72, 16, 3328),
    "render_streams_tree_kernel<true, 8>": (96, 16 * 14 * 4 + 16, 5632),   # 5 waves/SIMD; the lane stack IS scratch: 16 entries x 14 words
    "streams_level_kernel<true, true, true>": (80, 0, 4608),     # 6 waves/SIMD, nothing in scratch
    "streams_level_kernel<true, true, false>": (80, 0, 0),
    "streams_level_kernel<true, false, false>": (80, 0, 0),
}


@pytest.mark.parametrize("kernel", sorted(BUDGETS))
def test_render_kernel_stays_within_its_budget(resources, kernel):
    assert kernel in resources, sorted(resources)
    vgpr, scratch, lds = BUDGETS[kernel]
    r = resources[kernel]
    assert r["vgpr"] <= vgpr, r
    assert r["scratch"] <= scratch, r
    assert r["lds"] <= lds, r


def test_no_render_loop_touches_scratch_except_the_lane_stack(resources):
    """Spills around a loop show up as a handful of scratch instructions; the per-pixel tree walk's stack as 8 loads / 4
    stores.  Anything more means registers or flags went to memory inside a loop."""
    for name, r in resources.items():
        loads, stores = (8 + 2, 4 + 2) if "tree" in name else (4, 4)
        if name.startswith(("render_inline_kernel<true, 0", "render_inline_kernel<false, 0", "render_streams_kernel",
                            "render_streams_tree_kernel", "streams_level_kernel")):
            assert r["scratch_loads"] <= loads and r["scratch_stores"] <= stores, (name, r)
