"""haskell/patches/*.diff apply to the reference tree (CPU; skipped where /root/reference does not exist, e.g. on the GPU box).
The patched tree is never compiled -- no GHC anywhere -- this only keeps the patches from rotting against the files they edit."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
PATCHES = os.path.join(ROOT, "haskell-path-tracer_amd", "haskell", "patches")
FILES = {"Main.hs.diff": "app/Main.hs", "World.hs.diff": "src/Scene/World.hs", "tracer.cabal.diff": "tracer.cabal"}
NEEDS_REFERENCE = pytest.mark.skipif(not os.path.isdir(REFERENCE) or shutil.which("patch") is None, reason="needs the reference tree and patch(1)")


@NEEDS_REFERENCE
def test_patches_apply_to_the_reference_tree(tmp_path):
    for rel in FILES.values():
        os.makedirs(os.path.dirname(os.path.join(tmp_path, rel)), exist_ok=True)
        shutil.copy(os.path.join(REFERENCE, rel), os.path.join(tmp_path, rel))
    for name, rel in FILES.items():
        dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(PATCHES, name)], cwd=tmp_path, capture_output=True, text=True)
        assert dry.returncode == 0, name + ": " + dry.stdout + dry.stderr
        real = subprocess.run(["patch", "-p1", "-i", os.path.join(PATCHES, name)], cwd=tmp_path, capture_output=True, text=True)
        assert real.returncode == 0 and "fuzz" not in real.stdout, name + ": " + real.stdout
    main = open(os.path.join(tmp_path, "app/Main.hs")).read()
    assert "USE_HIP_BACKEND" in main and "HIP.compileFor hip" in main and main.count("runInitialOutput") == 6     # two call sites, two definitions with their signatures
    assert "run <$> initialOutput" in main and "run <$> (reseed . A.use $ acc)" in main          # the other backends' branch is untouched
    world = open(os.path.join(tmp_path, "src/Scene/World.hs")).read()
    assert "mainScene' :: ([Sphere], [Plane])" in world
    cabal = open(os.path.join(tmp_path, "tracer.cabal")).read()
    assert "extra-libraries: ptmi" in cabal and "flag Hip" in cabal and "-DUSE_HIP_BACKEND" in cabal


@NEEDS_REFERENCE
def test_the_resident_wiring_applies_to_a_pristine_main(tmp_path):
    """patches/Main.resident.diff -- the alternative to Main.hs.diff: Result carries the iteration count, the batch is
    one HIP.renderResident, the colour planes come down under the lock.  Applies to the reference's app/Main.hs without fuzz; every
    Accelerate line it touches is kept under #else."""
    os.makedirs(tmp_path / "app")
    shutil.copy(os.path.join(REFERENCE, "app/Main.hs"), tmp_path / "app" / "Main.hs")
    patch = os.path.join(PATCHES, "Main.resident.diff")
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert dry.returncode == 0, dry.stdout + dry.stderr
    real = subprocess.run(["patch", "-p1", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert real.returncode == 0 and "fuzz" not in real.stdout and "offset" not in real.stdout, real.stdout
    main = open(tmp_path / "app" / "Main.hs").read()
    original = open(os.path.join(REFERENCE, "app/Main.hs")).read()
    assert main.count("#if defined(USE_HIP_BACKEND)") == main.count("#endif") == 9           # (the reference's own backend #ifdef is now an #elif of the first)
    for call in ("HIP.renderResident hip", "HIP.resetOutput hip", "HIP.reseedResident hip", "HIP.downloadColor hip", "HIP.synchronize hip", "HIP.initialise 0 screenWidth screenHeight"):
        assert call in main, call
    assert "type CompiledFunction = Camera -> Int -> Int -> IO Int" in main and "type Accumulator = ()" in main and "(Int, Accumulator)" in main
    # with the HIP branches cut out, what is left is the reference's file but for the Accumulator alias
    import re
    stripped = re.sub(r"#if defined\(USE_HIP_BACKEND\)\n.*?#elif defined\(USE_CPU_BACKEND\)\n", "#ifdef USE_CPU_BACKEND\n", main, count=1, flags=re.S)
    stripped = re.sub(r"#if defined\(USE_HIP_BACKEND\)\n.*?#else\n(.*?)#endif\n", r"\1", stripped, flags=re.S)
    stripped = stripped.replace("type Accumulator = RenderResult\n", "").replace("(Int, Accumulator)", "(Int, RenderResult)")
    assert stripped == original
    src = open(os.path.join(ROOT, "haskell-path-tracer_amd", "haskell", "Scene", "HIP.hs")).read()
    head = src[src.index("module Scene.HIP"):src.index(") where")]
    for name in ("initialise", "resetOutput", "renderResident", "reseedResident", "downloadColor", "synchronize"):
        assert name in head


def test_the_module_the_patches_import_exports_what_they_use():
    """Scene/HIP.hs (source only) exports the names Main.hs.diff calls."""
    src = open(os.path.join(ROOT, "haskell-path-tracer_amd", "haskell", "Scene", "HIP.hs")).read()
    head = src[src.index("module Scene.HIP"):src.index(") where")]
    for name in ("Handle", "initialise", "compileFor", "initialOutput", "reseed"):
        assert name in head
    patch = open(os.path.join(PATCHES, "Main.hs.diff")).read()
    for call in ("HIP.initialise", "HIP.compileFor", "HIP.initialOutput", "HIP.reseed"):
        assert call in patch
