"""The multi-device entry of the C ABI (ptmi_group_*): partition arithmetic on the CPU; on the GPU a group's stitched
read-out -- host planes and RCCL gather to a root device -- equals the ungrouped image bit for bit."""

import numpy as np
import pytest

from conftest import assert_planes_equal


def test_partition_arithmetic_matches_the_python_mirror(pkg):
    """ptmi_partition_rows / ptmi_partition_global_row (no device needed) against parallel.StripePartition, and the
    parts of any partition tile the image exactly once."""
    from haskell_path_tracer_amd.parallel import StripePartition
    lib = pkg.load_library()
    for height, stripe, n_parts in [(2160, 10, 8), (2160, 8, 8), (1080, 8, 3), (67, 8, 5), (7, 8, 2), (600, 1, 4), (33, 5, 7)]:
        seen = np.zeros(height, np.int32)
        for part in range(n_parts):
            p = StripePartition(height, n_parts, part, stripe)
            rows = lib.ptmi_partition_rows(height, stripe, n_parts, part)
            assert rows == p.local_rows
            got = np.array([lib.ptmi_partition_global_row(height, stripe, n_parts, part, i) for i in range(rows)], np.int64)
            assert np.array_equal(got, p.global_rows())
            seen[got] += 1
            assert lib.ptmi_partition_global_row(height, stripe, n_parts, part, rows) == -1
        assert np.all(seen == 1)
    assert lib.ptmi_partition_rows(0, 8, 2, 0) == -1 and lib.ptmi_partition_rows(10, 8, 2, 2) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("devices,stripe", [([0], 0), ([0, 0], 8), ([0, 0, 0], 5)])
def test_group_host_readout_equals_the_ungrouped_image(pkg, devices, stripe):
    """A group of 1 member, and groups of 2 and 3 members that share the one GPU of the test box (each member is a
    context of its own with its own partition, as on separate devices)."""
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h = 333, 131
    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        c.resize(w, h)
        c.init_output(7)
        c.render(cam, 8, 3)
        want = c.download_state()
        live = c.stats()["live_bounces"]
    with pkg.Group(devices, stripe) as g:
        assert g.size == len(devices)
        g.set_scene(sp, pl)
        g.resize(w, h)
        g.init_output(7)
        g.render(cam, 8, 2)
        g.render(cam, 8, 1)
        g.synchronize()
        got = g.download_color()
        st = g.stats()
        assert sum(g.member(i).local_rows for i in range(g.size)) == h
    assert_planes_equal(got, want[:3], "group of %d, host read-out" % len(devices))
    assert st["live_bounces"] == live and st["samples"] == w * h * 3


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [False, True])
def test_group_device_gather_one_member(pkg, monkeypatch, force_rccl):
    """ptmi_group_gather_color on a 1-member group: with PTMI_GROUP_FORCE_RCCL=1 the planes really travel through
    ncclSend / ncclRecv (to self) on a communicator made by ncclCommInitAll, then through the stitch kernel."""
    if force_rccl:
        monkeypatch.setenv("PTMI_GROUP_FORCE_RCCL", "1")
    sp, pl = pkg.world.main_scene()
    cam = pkg.world.initial_camera()
    w, h = 200, 77
    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        c.resize(w, h)
        c.init_output(3)
        c.render(cam, 15, 2)
        want = c.download_color()
    with pkg.Group([0]) as g:
        g.set_scene(sp, pl)
        g.resize(w, h)
        g.init_output(3)
        g.render(cam, 15, 2)
        with pkg.Context(0) as out:                      # a second context lends its colour planes as the [h][w] destination
            out.resize(w, h)
            out.upload_state(*[np.full((h, w), -1.0, np.float32)] * 3)
            r, gp, b = out.device_planes()[:3]
            g.gather_color(0, r, gp, b)
            got = out.download_color()
    assert_planes_equal(got, want, "device gather (%s)" % ("RCCL" if force_rccl else "copy"))


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["s16", "glass"])
def test_group_renders_the_stream_form_on_all_members(pkg, ora, scene_name):
    """render Streams in its stream form on every member of a group (ptmi_group_set_option).  Without a ray-splitting material a
    member only enqueues and the stitched colours equal the ungrouped per-pixel Streams image bit for bit; with GLASS every call
    reads its overflow counters back (ptmi_render_blocks), so ptmi_group_render gives the members a host thread each, and the
    stitched colours equal the ungrouped stream-form image within the tolerance of the undefined addition order."""
    B = pkg.binding
    sp, pl = pkg.world.scene16() if scene_name == "s16" else pkg.world.glass_scene()
    cam = pkg.world.initial_camera()
    w, h = 211, 97
    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        c.resize(w, h)
        c.init_output(11)
        c.render(cam, 8, 3, pkg.STREAMS)
        want = c.download_color()
        live = c.stats()["live_bounces"]
        assert not c.render_blocks(pkg.STREAMS)
        c.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        assert c.render_blocks(pkg.STREAMS) == (scene_name == "glass") and not c.render_blocks(pkg.INLINE)
    with pkg.Group([0, 0, 0], 4) as g:
        g.set_scene(sp, pl)
        g.resize(w, h)
        g.init_output(11)
        g.set_option(B.OPT_STREAMS_FORM, B.FORM_STREAM)
        assert all(g.member(i).get_option(B.OPT_STREAMS_FORM) == B.FORM_STREAM for i in range(g.size))
        g.render(cam, 8, 3, pkg.STREAMS)
        g.synchronize()
        got = g.download_color()
        st = g.stats()
        assert st["live_bounces"] == live and st["stream_rays_dropped"] == 0
        with pytest.raises(pkg.PtmiError):
            g.set_option(99, 0)
        g.set_variant(0)
    if scene_name == "s16":
        assert_planes_equal(got, want, "group of 3, stream form")
    else:
        for a, b in zip(got, want):
            assert np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)) <= 1e-4


@pytest.mark.gpu
def test_group_of_8_in_the_8_gpu_shape(pkg):
    """The shape of the 8-GPU job in ONE process: a 3840x2160 image, 10-row stripes over a group of 8 members (270 rows each;
    here they share the one device of the test box).  ptmi_group_download_color equals the whole image of one context bit for
    bit, and so do the members' live-bounce counts in sum."""
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h, spp = 3840, 2160, 2
    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        c.resize(w, h)
        c.init_output(0x5EED1234)
        c.render(cam, 8, spp)
        want = c.download_color()
        live = c.stats()["live_bounces"]
    with pkg.Group([0] * 8, 10) as g:
        g.set_scene(sp, pl)
        g.resize(w, h)
        assert [g.member(i).local_rows for i in range(8)] == [270] * 8
        g.init_output(0x5EED1234)
        g.render(cam, 8, spp)
        got = g.download_color()
        st = g.stats()
    assert_planes_equal(got, want, "group of 8, 10-row stripes, 4K")
    assert st["live_bounces"] == live and st["samples"] == w * h * spp


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.skipif(_device_count() < 2, reason="needs two physical GPUs: arms itself on the first multi-GPU box the suite meets")
@pytest.mark.parametrize("root", [0, 1])
def test_group_gathers_between_two_physical_devices(pkg, root):
    """ptmi_group_gather_color between REAL devices (ptmi_group.cpp: ncclCommInitAll over the members' devices, grouped ncclSend /
    ncclRecv on per-member streams over xGMI, the stitch kernel on the root) -- never run before round 4's boxes: every earlier test shared
    one device.  As many members as the box has devices (at most 8), 4K in 10-row stripes; the gathered planes on either root, and the
    host read-out, equal one context's image bit for bit."""
    n = min(_device_count(), 8)
    sp, pl = pkg.world.scene16()
    cam = pkg.world.initial_camera()
    w, h, spp = 3840, 2160, 2
    with pkg.Context(0) as c:
        c.set_scene(sp, pl)
        c.resize(w, h)
        c.init_output(0x5EED1234)
        c.render(cam, 8, spp)
        want = c.download_color()
        live = c.stats()["live_bounces"]
    with pkg.Group(list(range(n)), 10) as g:
        g.set_scene(sp, pl)
        g.resize(w, h)
        g.init_output(0x5EED1234)
        g.render(cam, 8, spp)
        g.synchronize()
        assert_planes_equal(g.download_color(), want, "group of %d devices, host read-out" % n)
        assert g.stats()["live_bounces"] == live
        with pkg.Context(root) as out:                  # the destination planes live on the root member's device
            out.resize(w, h)
            out.upload_state(*[np.full((h, w), -1.0, np.float32)] * 3)
            r, gp, b = out.device_planes()[:3]
            for _ in range(2):                           # the second gather reuses the communicator
                g.gather_color(root, r, gp, b)
            got = out.download_color()
    assert_planes_equal(got, want, "group of %d devices, RCCL gather to member %d" % (n, root))
