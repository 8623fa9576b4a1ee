"""tools/compare_ghc_dump.py, the MI355X side of the parity experiment, fed dumps in the format haskell/dump/Dump.hs writes -- made here
FROM THE ORACLE (a real one needs a GHC build of the reference, which does not exist in this container): a faithful dump must pass every
assumption; one with a deliberately broken generator must be named by its assumption.  Also: the numpy SFC32 the comparator diagnoses
with equals the oracle's, the file format round-trips, and the Haskell program's cabal patch applies to the reference tree."""
import importlib.util
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP_DIR = os.path.join(ROOT, "haskell-path-tracer_amd", "haskell", "dump")


@pytest.fixture(scope="module")
def tool():
    spec = importlib.util.spec_from_file_location("compare_ghc_dump", os.path.join(ROOT, "tools", "compare_ghc_dump.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_comparators_numpy_generator_is_the_oracles(tool, ora):
    r = np.random.default_rng(5)
    w = r.integers(0, 2 ** 32, (3, 40), dtype=np.uint64).astype(np.uint32)
    state = tool.sfc32_seed3(w[0], w[1], w[2])
    for k in range(40):
        want = ora.sfc32_seed3(int(w[0][k]), int(w[1][k]), int(w[2][k]))
        assert tuple(int(p[k]) for p in state) == tuple(want)
        words, floats, _ = ora.sfc32_stream(want, 6)
        s = tuple(p[k:k + 1] for p in state)
        for j in range(6):
            out, s = tool.sfc32_next(s)
            assert int(out[0]) == int(words[j])
            assert tool.word_to_float(out)[0].view(np.uint32) == np.float32(floats[j]).view(np.uint32)
    edge = np.array([0, 1, 0x7fffffff, 0x80000000, 0xffffffff], np.uint32)       # (0, 1]: the largest word's neighbour gives 1.0, the int32 minimum the smallest
    f = tool.word_to_float(edge)
    assert f.min() > 0.0 and f.max() <= 1.0


def test_a_faithful_dump_passes_every_assumption(tool, tmp_path):
    path = str(tmp_path / "faithful.bin")
    tool.synthesize(path, 96, 64)
    dump = tool.read_dump(path)
    assert (dump["width"], dump["height"], dump["limit"]) == (96, 64, 15)
    assert [len(dump["sections"][k]) for k in ("words", "created", "probe_states", "probe_words", "probe_floats", "inline_1", "streams_2")] == [3, 4, 4, 4, 4, 7, 7]
    report = tool.analyse(dump, use_device=False)
    assert report["all_pass"] and report["first_failure"] is None, report["assumptions"]
    assert report["arithmetic"] is None                       # (which arithmetic the dump is closer to is asked on a GPU box only)
    assert report["assumptions"]["A3"]["plane_order_in_toVectors"] == ["a", "b", "c", "counter"]
    for tag in ("inline", "streams_from_result"):
        for k in ("after_1", "after_2"):
            m = report["renders"][tag][k]
            assert m["colour_bit_identical"] == 1.0 and m["rng_state_identical"] == 1.0 and m["first_differing_pixel_y_x"] is None
    assert report["renders"]["streams_keep_accumulator"]["after_1"]["rng_state_identical"] < 0.9       # the other reading of `combine` is told apart


@pytest.mark.parametrize("perturb", ["A1", "A2", "A3", "A4", "A5"])
def test_a_broken_assumption_is_named(tool, tmp_path, perturb):
    path = str(tmp_path / ("broken_%s.bin" % perturb))
    tool.synthesize(path, 64, 40, perturb)
    report = tool.analyse(tool.read_dump(path), use_device=False)
    assert not report["all_pass"]
    assert report["first_failure"] == perturb, report["assumptions"]
    A = report["assumptions"]
    if perturb == "A1":                                       # rotl by 20 instead of 21: seeding (A2) cannot be judged, and the hint finds the rotation
        assert A["A2"]["status"] == "undetermined" and "20" in (A["A1"]["hint"] or "")
    if perturb == "A2":
        assert A["A1"]["status"] == "pass" and "12 outputs discarded" in A["A2"]["hint"]
    if perturb == "A3":                                       # planes (b, a, c, counter): everything else still holds, and the report says which order
        assert A["A1"]["status"] == A["A2"]["status"] == A["A4"]["status"] == A["A6"]["status"] == "pass"
        assert A["A3"]["plane_order_in_toVectors"] == ["b", "a", "c", "counter"]
    if perturb == "A4":
        assert A["A1"]["status"] == "pass" and "top24" in A["A4"]["hint"]
    if perturb == "A5":
        assert A["A6"]["status"] == "pass" and "KEEP_ACCUMULATOR" in A["A5"]["detail"]


def test_command_line_and_exit_codes(tmp_path):
    tool_py = os.path.join(ROOT, "tools", "compare_ghc_dump.py")
    good, bad = str(tmp_path / "good.bin"), str(tmp_path / "bad.bin")
    import sys
    assert subprocess.run([sys.executable, tool_py, "--synthesize", good, "--size", "48x32"]).returncode == 0
    assert subprocess.run([sys.executable, tool_py, "--synthesize", bad, "--size", "48x32", "--perturb", "A1"]).returncode == 0
    ok = subprocess.run([sys.executable, tool_py, good, "--no-device", "--json", str(tmp_path / "r.json")], capture_output=True, text=True)
    assert ok.returncode == 0 and "all pass: True" in ok.stdout and os.path.exists(tmp_path / "r.json")
    ko = subprocess.run([sys.executable, tool_py, bad, "--no-device"], capture_output=True, text=True)
    assert ko.returncode == 1 and "first failure: A1" in ko.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference") or shutil.which("patch") is None, reason="needs the reference tree and patch(1)")
def test_the_dump_programs_cabal_patch_applies_and_its_source_names_what_the_reader_expects(tmp_path):
    shutil.copy("/root/reference/tracer.cabal", tmp_path / "tracer.cabal")
    res = subprocess.run(["patch", "-p1", "-i", os.path.join(DUMP_DIR, "tracer.cabal.diff")], cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0 and "fuzz" not in res.stdout, res.stdout + res.stderr
    cabal = open(tmp_path / "tracer.cabal").read()
    assert "executable ptmi-dump" in cabal and "accelerate-llvm-native" in cabal.split("executable ptmi-dump")[1]
    # ... also on top of the wiring patch of haskell/patches
    shutil.copy("/root/reference/tracer.cabal", tmp_path / "tracer.cabal")
    for diff in (os.path.join(ROOT, "haskell-path-tracer_amd", "haskell", "patches", "tracer.cabal.diff"), os.path.join(DUMP_DIR, "tracer.cabal.diff")):
        assert subprocess.run(["patch", "-p1", "-i", diff], cwd=tmp_path, capture_output=True, text=True).returncode == 0
    src = open(os.path.join(DUMP_DIR, "Dump.hs")).read()
    for name in ("words", "created", "probe_states", "probe_words", "probe_floats", "inline_1", "inline_2", "streams_1", "streams_2"):
        assert 'section "%s"' % name in src
    assert '"PTMIDUMP"' in src and "word32LE 1 " in src and "runN (render algorithm) screenPixels" in src
    for imported in ("Scene.Trace", "Scene.World", "Util", "Data.Array.Accelerate.System.Random.SFC", "Data.Array.Accelerate.LLVM.Native"):
        assert "import           " + imported in src or "import qualified " + imported in src


@pytest.mark.gpu
def test_the_comparator_drives_libptmi_on_the_dumps_inputs(tool, tmp_path):
    """With a GPU the comparator also renders the dump's inputs through libptmi (upload_state of the created planes, one sample per call,
    800x600 / limit 15 as Dump.hs writes them): the device must equal the oracle on them, for Inline and for Streams under both seed rules."""
    path = str(tmp_path / "c0.bin")
    tool.synthesize(path, 800, 600)
    report = tool.analyse(tool.read_dump(path), use_device=True)
    assert report["all_pass"]
    for tag in ("inline", "streams_from_result", "streams_keep_accumulator"):
        assert report["renders"][tag].get("device_equals_oracle") is True, report["renders"][tag]
    # ... and asks which arithmetic the dump's is: a dump made by the oracle is identical to libptmi's exact arithmetic and not to the contracted one
    ar = report["arithmetic"]
    assert ar["closer_to_the_dump"] == "exact" and ar["identical_to_the_dump"] == ["exact"]
    assert ar["contracted"]["after_2"]["colour_bit_identical"] < 1.0 and ar["contracted"]["after_2"]["colour_within_1e-4"] > 0.99


@pytest.mark.gpu
def test_a_dump_from_a_contracting_backend_is_recognised_as_such(tool, tmp_path):
    """--perturb A6: the Inline planes as a backend that fuses a * b + c would have written them (libptmi's PTMI_ARITH_CONTRACTED, the only
    contracted evaluation this repository has).  The generator checks pass, A6 / A7 fail as "rounding only", and the arithmetic question says
    `contracted` -- the answer about A6 that only a real dump can give."""
    path = str(tmp_path / "contracted.bin")
    tool.synthesize(path, 800, 600, "A6")
    report = tool.analyse(tool.read_dump(path), use_device=True)
    A = report["assumptions"]
    assert all(A[k]["status"] == "pass" for k in ("A1", "A2", "A3", "A4", "A5"))
    assert report["first_failure"] == "A6" and "rounding only" in A["A6"]["detail"]
    ar = report["arithmetic"]
    assert ar["closer_to_the_dump"] == "contracted" and ar["identical_to_the_dump"] == ["contracted"]
    assert ar["exact"]["after_2"]["colour_within_1e-4"] > 0.99
