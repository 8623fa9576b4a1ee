"""The C-ABI library loads and exports every symbol include/ptmi.h declares (no compute calls
without a GPU); the binding, the header and the library agree on the symbol set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptmi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for must in ("ptmi_create", "ptmi_destroy", "ptmi_set_scene", "ptmi_resize", "ptmi_render", "ptmi_render1",
                 "ptmi_init_output", "ptmi_reseed", "ptmi_download_color", "ptmi_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg._build.LIB)
    for name in header_symbols():
        assert hasattr(lib, name), "libptmi.so does not export %s" % name


def test_binding_covers_exactly_the_header(pkg):
    assert sorted(pkg.SYMBOLS) == header_symbols()
    pkg.load_library()          # types every symbol; AttributeError if one is missing


def test_binding_constants_equal_the_headers_enumerators(pkg):
    """Every PTMI_OPT_* / PTMI_FORM_* / PTMI_SEED_* / PTMI_ARITH_* / error enumerator of include/ptmi.h that the binding names has the header's value."""
    text = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    values = {name: int(v, 0) for name, v in re.findall(r"\b(PTMI_[A-Z0-9_]+)\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+))", text)}
    B = pkg.binding
    options = {n: v for n, v in values.items() if n.startswith("PTMI_OPT_")}
    assert len(options) >= 12
    checked = 0
    for name, value in values.items():
        for prefix in ("PTMI_OPT_", "PTMI_FORM_", "PTMI_ARITH_"):
            if name.startswith(prefix):
                short = name[len("PTMI_"):]
                assert hasattr(B, short), "binding.py has no %s" % short
                assert getattr(B, short) == value, name
                checked += 1
        if hasattr(B, name):                                   # error codes and the like keep their full names
            assert getattr(B, name) == value, name
            checked += 1
    assert checked >= 20


def test_version_and_strerror(pkg):
    lib = pkg.load_library()
    assert lib.ptmi_version() == 500
    assert lib.ptmi_strerror(0) == b"ok" and lib.ptmi_strerror(-2) == b"no usable HIP device"


def test_library_carries_the_hash_of_the_sources_it_was_built_from(pkg):
    """ptmi_build_id() of the loaded library == the id read from the file == _build.source_hash() of the sources beside it."""
    lib = pkg.load_library()
    want = pkg._build.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    assert lib.ptmi_build_id().decode() == want == lib.build_id
    assert pkg._build.read_build_id(pkg._build.LIB) == want
    assert not pkg._build.is_stale()
    assert pkg._build.build_id(["-DPTMI_ABLATIONS"]) == want + "+PTMI_ABLATIONS"


def test_a_library_built_from_other_sources_is_refused(pkg, tmp_path):
    """A binary whose id is not the hash of the sources here (it travelled with edited sources, or comes from another checkout)
    is refused by the binding and counts as stale for the build, whatever its file time says."""
    import shutil
    want = pkg._build.source_hash().encode()
    blob = open(pkg._build.LIB, "rb").read()
    marker = pkg._build.BUILD_ID_MARKER + want
    assert blob.count(marker) == 1
    foreign = bytes(reversed(want)) if bytes(reversed(want)) != want else b"0" * 16
    other = tmp_path / "libptmi_foreign.so"
    other.write_bytes(blob.replace(marker, pkg._build.BUILD_ID_MARKER + foreign))
    os.utime(other, (2 ** 31, 2 ** 31))                      # newer than every source: an mtime rule would call it current
    assert pkg._build.read_build_id(str(other)) == foreign.decode()
    assert pkg._build.is_stale(str(other))
    with pytest.raises(pkg.PtmiError) as e:
        pkg.binding.open_library(str(other))
    assert e.value.code == pkg.binding.PTMI_ESTATE and foreign.decode() in str(e.value)
    lib = pkg.binding.open_library(str(other), check_build_id=False)       # (the explicit way round the check, for archaeology)
    assert lib.build_id == foreign.decode()
    assert pkg._build.read_build_id(str(tmp_path / "missing.so")) is None
    shutil.copy(__file__, tmp_path / "not_a_library.so")
    assert pkg._build.read_build_id(str(tmp_path / "not_a_library.so")) is None


def test_debug_counters_capacity(pkg):
    """ptmi_debug_counters keeps its 64-word contract; the sized entry point is declared with a capacity."""
    text = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    assert "int ptmi_debug_counters(ptmi_ctx *ctx, uint32_t out[64]);" in text
    assert "int ptmi_debug_counters_n(ptmi_ctx *ctx, uint32_t *out, int capacity);" in text
    assert "getenv(\"PTMI_ORDERED_PASSES\")" not in open(os.path.join(ROOT, "haskell-path-tracer_amd", "csrc", "ptmi_api.cpp")).read()


def test_struct_layouts_match_header(pkg):
    w = pkg.world
    assert w.SPHERE_DTYPE.itemsize == 40 and w.PLANE_DTYPE.itemsize == 48 and w.CAMERA_DTYPE.itemsize == 32
    assert w.CAMERA_DTYPE.fields["fov"][1] == 24 and w.PLANE_DTYPE.fields["direction"][1] == 12
    assert w.SPHERE_DTYPE.fields["brdf_tag"][1] == 32


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    """Without a device the product refuses to work (PTMI_ENODEVICE); it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PtmiError) as e:
        pkg.Context(0)
    assert e.value.code == -2


def test_product_does_not_reference_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import, link or call it."""
    pkg_dir = os.path.join(ROOT, "haskell-path-tracer_amd")
    for dirpath, _dirs, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".hs")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "pt_oracle" not in text and "libptoracle" not in text, f
