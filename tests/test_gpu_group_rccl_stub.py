"""ptmi_group_gather_color's n > 1 branch, EXECUTED: csrc/ptmi_group.cpp:326-393 -- ncclCommInitAll over the group's
devices, grouped ncclSend / ncclRecv on per-member streams, offsets into the root's receive block, the stitch kernel -- had only ever run with
one member sending to itself, because every box this build has seen holds one GPU and real RCCL refuses a communicator with duplicate devices.

A SUBPROCESS puts a test-only stand-in for librccl.so.1 (tests/cxx/rccl_stub.cpp: paired sends and receives become hipMemcpyAsync on the
given streams) first on its loader path, so that groups of 3 and 8 members sharing device 0 go through the real branch.  This validates the
HOST LOGIC only -- offsets, counts, roots other than 0, unequal row counts, members without rows, communicator reuse, an error inside the
group -- not RCCL, not xGMI.  tests/test_group.py::test_group_gathers_between_two_physical_devices stays armed for the first multi-GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_DIR = os.path.join(ROOT, "build", "rccl_stub")
STUB = os.path.join(STUB_DIR, "librccl.so.1")
SRC = os.path.join(ROOT, "tests", "cxx", "rccl_stub.cpp")


def build_stub():
    os.makedirs(STUB_DIR, exist_ok=True)
    if os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(SRC):
        return STUB
    hipcc = "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wl,-soname,librccl.so.1", SRC, "-o", STUB]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    return STUB


def test_the_stub_builds_and_exports_what_the_group_resolves():
    """CPU: the stand-in compiles and exports the seven entry points ptmi_group.cpp looks up with dlsym."""
    # (read with nm, not loaded: a library of that soname inside THIS process would be handed to torch when it asks for RCCL)
    out = subprocess.run(["nm", "-D", "--defined-only", build_stub()], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    exported = {line.split()[-1] for line in out.stdout.splitlines() if line.strip()}
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString", "rccl_stub_counters"):
        assert name in exported, name
    src = open(os.path.join(ROOT, "haskell-path-tracer_amd", "csrc", "ptmi_group.cpp")).read()
    for name in ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString"):
        assert 'sym("%s")' % name in src


SCRIPT = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as graft
pkg = graft.load_package()
assert "torch" not in sys.modules                        # (torch would bring the real librccl into the process)
stub = ctypes.CDLL("librccl.so.1")                      # by NAME: what ptmi_group.cpp's dlopen will find too
assert stub.rccl_stub_is_the_stub() == 1

def counters():
    out = (ctypes.c_long * 8)()
    stub.rccl_stub_counters(out)
    return dict(zip("init_all comms sends recvs groups copies destroys bytes".split(), out))

sp, pl = pkg.world.scene16()
cam = pkg.world.initial_camera()
checked = []

def one(n, stripe, w, h, roots, spp=2, fail=False):
    with pkg.Context(0) as c:
        c.set_scene(sp, pl); c.resize(w, h); c.init_output(7); c.render(cam, 8, spp)
        want = c.download_color()
    before = counters()
    with pkg.Group([0] * n, stripe) as g:
        g.set_scene(sp, pl); g.resize(w, h); g.init_output(7); g.render(cam, 8, spp)
        rows = [g.member(i).local_rows for i in range(n)]
        assert sum(rows) == h
        with pkg.Context(0) as out:                      # lends its colour planes as the [h][w] destination on the root's device
            out.resize(w, h)
            for k, root in enumerate(roots):
                out.upload_state(*[np.full((h, w), -1.0, np.float32)] * 3)
                r, gp, b = out.device_planes()[:3]
                if fail and k == 0:
                    stub.rccl_stub_fail_send(2)             # the second ncclSend of this gather fails, between GroupStart and GroupEnd
                    try:
                        g.gather_color(root, r, gp, b)
                        raise SystemExit("the injected ncclSend error did not surface")
                    except pkg.PtmiError as e:
                        assert e.code == pkg.binding.PTMI_EHIP and "ncclSend" in str(e) and "injected" in str(e), str(e)
                    g.synchronize()
                g.gather_color(root, r, gp, b)              # (after a failed gather: the group was closed, the next one works)
                got = out.download_color()
                for name, a, bb in zip("rgb", got, want):
                    assert np.array_equal(a.view(np.uint32), bb.view(np.uint32)), ("plane %%s differs: n=%%d stripe=%%d %%dx%%d root=%%d" %% (name, n, stripe, w, h, root))
        host = g.download_color()
        for a, bb in zip(host, want):
            assert np.array_equal(a.view(np.uint32), bb.view(np.uint32))
    after = counters()
    d = {k: after[k] - before[k] for k in after}
    senders = [sum(1 for i in range(n) if i != root and rows[i] > 0) for root in roots]
    assert d["init_all"] == 1 and d["comms"] == n and d["destroys"] == n, d        # one communicator set per group, reused by every gather, destroyed with it
    assert d["groups"] == len(roots) + (1 if fail else 0), d
    if not fail:
        assert d["sends"] == d["recvs"] == d["copies"] == sum(senders), (d, senders)
        assert d["bytes"] == sum(12 * w * sum(rows[i] for i in range(n) if i != root) for root in roots), d
    checked.append("n=%%d stripe=%%d %%dx%%d rows=%%s roots=%%s%%s" %% (n, stripe, w, h, rows, roots, " +injected error" if fail else ""))

one(3, 5, 333, 131, [0, 2, 1, 2])                      # 131 rows in 5-row stripes over 3: 45 / 45 / 41 rows; every root; a root twice
one(8, 8, 640, 333, [0, 5, 7])                         # 8 members: 48 / 45 / 40 x 6 rows
one(8, 8, 96, 20, [0, 1, 6])                           # 20 rows: members 3..7 hold NOTHING (no send, no receive); root 6 is one of them
one(8, 10, 3840, 2160, [3], spp=1)                     # the 8-GPU job's shape: 4K, 10-row stripes, 270 rows each
one(3, 4, 211, 97, [1, 0], fail=True)                  # an error between ncclGroupStart and ncclGroupEnd, then a good gather
print("STUB_GATHER_OK", len(checked))
for line in checked:
    print(line)
'''


@pytest.mark.gpu
def test_gather_through_the_stub_with_3_and_8_members_on_one_device(pkg):
    build_stub()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = STUB_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env.pop("PTMI_GROUP_FORCE_RCCL", None)
    res = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "STUB_GATHER_OK 5" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
