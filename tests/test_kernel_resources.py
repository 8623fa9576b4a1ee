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


# kernel -> (VGPRs for its waves per SIMD, scratch bytes, static LDS bytes, scratch loads, scratch stores).  The few scratch
# instructions allowed are spills AROUND a render loop and around the rare call of the literal fold (check_hit_exact).
BUDGETS = {
    "render_inline_kernel<true, 8>": (72, 32, 2560, 4, 4),          # 7 waves/SIMD
    "render_inline_kernel<false, 8>": (72, 32, 2560, 6, 6),         # scene through scalar loads (big scenes)
    "render_streams_kernel<true, 8>": (72, 32, 2816, 3, 3),            # 7 waves/SIMD
    "render_streams_kernel<false, 8>": (72, 32, 2816, 3, 3),
    "render_streams_tree_kernel<true, 8>": (80, 12 * 14 * 4 + 64, 5120, 8, 12),   # 6 waves/SIMD; waiting children: the first 4 per lane as global 64-byte records, the rest in scratch (12 entries x 14 words); three values spilled around the shade
    "streams_pixels_kernel<true, false>": (72, 0, 3584, 0, 0),         # stream form, rays never split: 7 waves/SIMD, nothing in scratch
    "streams_pixels_kernel<true, true>": (72, 0, 3584, 0, 0),          # ... with ordered passes
    "streams_split_kernel<true, true>": (80, 0, 5568, 0, 0),           # stream form with the child ring: 6 waves/SIMD (registers and LDS), nothing in scratch
    "streams_level_kernel<true>": (80, 0, 0, 0, 0),                    # overflow levels: 6 waves/SIMD
}


@pytest.mark.parametrize("kernel", sorted(BUDGETS))
def test_render_kernel_stays_within_its_budget(resources, kernel):
    assert kernel in resources, sorted(resources)
    vgpr, scratch, lds, loads, stores = BUDGETS[kernel]
    r = resources[kernel]
    assert r["vgpr"] <= vgpr, r
    assert r["scratch"] <= scratch, r
    assert r["lds"] <= lds, r
    assert r["scratch_loads"] <= loads and r["scratch_stores"] <= stores, r
