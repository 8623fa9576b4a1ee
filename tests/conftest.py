import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = graft.load_package()
    p._build.build_lib()
    return p


@pytest.fixture(scope="session")
def ora():
    o = graft.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def ablations(pkg):
    """libptmi_ablations.so: libptmi built with -DPTMI_ABLATIONS (the slower loop shapes of DESIGN.md 5.2, kept as evidence;
    the product library does not contain them)."""
    return pkg.binding.open_library(pkg._build.build_ablations_lib())


@pytest.fixture(scope="session")
def actx(pkg, ablations):
    """One device context of the ablation library (tests that walk through ptmi_set_variant's ablation values)."""
    c = pkg.Context(0, library=ablations)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx(pkg):
    """One device context for the GPU tests; fails loudly when the HIP library or GPU is missing."""
    c = pkg.Context(0)
    yield c
    c.close()


def initial_planes(ora, width, height, seed0=0x5EED1234):
    """initialOutput: zero colour + deterministic genSeeds, as host arrays (from the ORACLE's seeding)."""
    seeds = ora.gen_seeds(seed0, 0, width * height)
    return [np.zeros((height, width), np.float32) for _ in range(3)] + [s.reshape(height, width) for s in seeds]


def assert_planes_equal(got, want, what=""):
    names = "r g b sfc_a sfc_b sfc_c sfc_counter".split()
    for name, a, b in zip(names, got, want):
        a = np.asarray(a).reshape(-1).view(np.uint32)
        b = np.asarray(b).reshape(-1).view(np.uint32)
        if not np.array_equal(a, b):
            bad = np.flatnonzero(a != b)
            raise AssertionError("%s plane %s: %d of %d elements differ bitwise (first at %d: %#x vs %#x)"
                                 % (what, name, bad.size, a.size, bad[0], a[bad[0]], b[bad[0]]))


def sfc32_advance(state, draws):
    """The four SFC32 planes after `draws` raw steps (numpy, vectorised): what `draws` updateSeeds leave (Trace.hs:190-191)."""
    a, b, c, ctr = [np.array(p, dtype=np.uint32, copy=True) for p in state]
    with np.errstate(over="ignore"):
        for _ in range(draws):
            tmp = a + b + ctr
            ctr = ctr + np.uint32(1)
            a = b ^ (b >> np.uint32(9))
            b = c + (c << np.uint32(3))
            c = ((c << np.uint32(21)) | (c >> np.uint32(11))) + tmp
    return a, b, c, ctr


def initial_rows(ora, width, rows, seed0=0x5EED1234):
    """initialOutput for the image rows `rows` only (a row-stripe part): seeds come from the GLOBAL pixel index."""
    planes = [np.zeros((len(rows), width), np.float32) for _ in range(3)] + [np.empty((len(rows), width), np.uint32) for _ in range(4)]
    for k, row in enumerate(rows):
        for plane, s in zip(planes[3:], ora.gen_seeds(seed0, int(row) * width, width)):
            plane[k] = s
    return planes
