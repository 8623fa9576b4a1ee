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
    assert lib.ptmi_version() == 600
    assert lib.ptmi_strerror(0) == b"ok" and lib.ptmi_strerror(-2) == b"no usable HIP device"


def test_library_carries_the_id_of_the_code_it_holds(pkg):
    """ptmi_build_id() of the loaded library == the id read from the file == _build.code_id(): the hash over the allocated sections
    of the objects the present sources compile to.  The text hash it was linked from rides along, for staleness only."""
    lib = pkg.load_library()
    want = pkg._build.code_id()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    assert lib.ptmi_build_id().decode() == want == lib.build_id
    assert pkg._build.read_build_id(pkg._build.LIB) == want
    # the text hash the library was linked from is the present one, or the text has moved since without moving the code (a comment, a probe
    # that is empty in the product): either way the library is current
    assert re.fullmatch(r"[0-9a-f]{16}", pkg._build.read_source_hash(pkg._build.LIB)) and pkg._build.source_hash() != want
    assert not pkg._build.is_stale() and pkg._build.matches_sources(pkg._build.LIB)
    assert pkg._build.read_build_id(pkg._build.build_ablations_lib()) == pkg._build.code_id(["-DPTMI_ABLATIONS"])
    assert pkg._build.code_id(["-DPTMI_ABLATIONS"]).endswith("+PTMI_ABLATIONS") and pkg._build.code_id(["-DPTMI_ABLATIONS"]).split("+")[0] != want


def _copy_of_the_build(pkg, tmp_path):
    """The package's build script, kernel sources, header and object cache under another root, imported as a module of its own."""
    import importlib.util
    import shutil
    root = tmp_path / "checkout"
    shutil.copytree(os.path.join(ROOT, "haskell-path-tracer_amd", "csrc"), root / "haskell-path-tracer_amd" / "csrc")
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    shutil.copy(os.path.join(ROOT, "haskell-path-tracer_amd", "_build.py"), root / "haskell-path-tracer_amd" / "_build.py")
    shutil.copytree(pkg._build.OBJ_ROOT, root / "build" / "obj")        # (digests are over contents and repository-relative names: valid in the copy)
    shutil.copy(pkg._build.LIB, root / "haskell-path-tracer_amd" / "libptmi.so")
    spec = importlib.util.spec_from_file_location("ptmi_build_copy", str(root / "haskell-path-tracer_amd" / "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return root, mod


def test_a_comment_edit_keeps_the_build_id_and_a_code_edit_does_not(pkg, tmp_path):
    """the id names the CODE.  In a copy of the checkout (another path: no path is in an object), a comment added
    to ptmi_device.h -- every kernel unit includes it and is recompiled, every line of it moves down -- leaves code_id() and the
    library's standing untouched; one changed constant in one unit changes the id and makes the library stale."""
    root, b = _copy_of_the_build(pkg, tmp_path)
    lib = str(root / "haskell-path-tracer_amd" / "libptmi.so")
    want = pkg._build.code_id()
    assert b.code_id() == want and not b.is_stale(lib)                       # the copy compiles nothing: its cache came along
    device_h = root / "haskell-path-tracer_amd" / "csrc" / "ptmi_device.h"
    text = device_h.read_text()
    device_h.write_text("// a comment that was not here before,\n/* and a second\n   one */\n" + text.replace("// ", "//  ", 3))
    small = root / "haskell-path-tracer_amd" / "csrc" / "ptmi_small.hip"
    assert b.source_hash() != pkg._build.source_hash()                        # the TEXT moved ...
    assert b.code_id() == want                                                # ... the code did not (eight units recompiled to find out)
    assert not b.is_stale(lib) and b.matches_sources(lib)
    src = small.read_text()
    assert "255.0f" in src
    small.write_text(src.replace("255.0f", "254.0f", 1))                       # ptmi_present's scale: one constant of one kernel
    changed = b.code_id()
    assert changed != want and re.fullmatch(r"[0-9a-f]{16}", changed)
    assert b.is_stale(lib) and not b.matches_sources(lib)


def test_a_library_built_from_other_sources_is_refused(pkg, tmp_path):
    """A binary that holds other code than the sources here compile to (it travelled with edited kernels, or comes from another
    checkout) is refused by the binding and counts as stale for the build, whatever its file time says."""
    import shutil
    want = pkg._build.code_id().encode()
    text = pkg._build.read_source_hash(pkg._build.LIB).encode()      # (the text it was linked from: the present one, or an older one of the same code)
    blob = open(pkg._build.LIB, "rb").read()
    marker, text_marker = pkg._build.BUILD_ID_MARKER + want, pkg._build.SOURCE_HASH_MARKER + text
    assert blob.count(marker) == 1 and blob.count(text_marker) == 1
    foreign = bytes(reversed(want)) if bytes(reversed(want)) != want else b"0" * 16
    other = tmp_path / "libptmi_foreign.so"
    other.write_bytes(blob.replace(marker, pkg._build.BUILD_ID_MARKER + foreign).replace(text_marker, pkg._build.SOURCE_HASH_MARKER + b"f" * 16))
    os.utime(other, (2 ** 31, 2 ** 31))                      # newer than every source: an mtime rule would call it current
    assert pkg._build.read_build_id(str(other)) == foreign.decode()
    assert pkg._build.is_stale(str(other)) and not pkg._build.matches_sources(str(other))
    with pytest.raises(pkg.PtmiError) as e:
        pkg.binding.open_library(str(other))
    assert e.value.code == pkg.binding.PTMI_ESTATE and foreign.decode() in str(e.value)
    lib = pkg.binding.open_library(str(other), check_build_id=False)       # (the explicit way round the check, for archaeology)
    assert lib.build_id == foreign.decode()
    same_code = tmp_path / "libptmi_edited_comments.so"                      # the right code under a stale text hash: accepted (a comment was edited)
    same_code.write_bytes(blob.replace(text_marker, pkg._build.SOURCE_HASH_MARKER + b"e" * 16))
    assert not pkg._build.is_stale(str(same_code))
    assert pkg.binding.open_library(str(same_code)).build_id == want.decode()
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
