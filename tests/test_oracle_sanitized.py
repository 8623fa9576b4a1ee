"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers are not available on the pool): the C restatement
every parity claim rests on is compiled with -fsanitize=address,undefined and driven, in a child process, through every render entry point the
tests use -- render Inline, render Streams under both seed rules, the tree walk and the stream order with GLASS, rows and threads, the point
queries -- on the default scene, the glass scene and a scene sprinkled with zeros, denormals, huge values, infinities and NaNs.  Any report
(out-of-bounds access, use of an uninitialised or freed object, signed overflow, a bad shift, a float cast out of range) fails the test.
IEEE division by zero is NOT undefined here and is part of the contract (Intersection.hs:57-62: a ray parallel to a plane divides by 0)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as graft
pkg, ora = graft.load_package(), graft.load_oracle()
assert "sanitized" in ora.LIB
cam = pkg.world.initial_camera()
w, h = 40, 24
def start(seed):
    s = ora.gen_seeds(seed, 0, w * h)
    return [np.zeros((h, w), np.float32) for _ in range(3)] + [p.reshape(h, w) for p in s]
r = np.random.default_rng(11)
scenes = [pkg.world.main_scene(), pkg.world.scene16(), pkg.world.glass_scene(), pkg.world.mirror_box()]
sp, pl = pkg.world.glass_scene()
sp, pl = sp.copy(), pl.copy()
bad = np.array([0.0, -0.0, 1e-45, -1e-40, 1e30, -3e38, np.inf, -np.inf, np.nan], np.float32)
for arr, fields in ((sp, ("position", "radius", "color", "illuminance", "brdf_param")), (pl, ("position", "direction", "color", "illuminance", "brdf_param"))):
    for f in fields:
        flat = arr[f].reshape(-1)
        for i in range(flat.size):
            if r.random() < 0.15:
                flat[i] = bad[r.integers(0, bad.size)]
scenes.append((sp, pl))
with np.errstate(all="ignore"):
    for k, (spheres, planes) in enumerate(scenes):
        glass = bool((spheres["brdf_tag"] == 2).any() or (planes["brdf_tag"] == 2).any())
        if not glass:
            ora.render_inline(spheres, planes, cam, w, h, 15, 2, start(k), n_threads=2)
            ora.render_inline(spheres, planes, cam, w, h, 0, 1, start(k))
            rows = np.array([1, 5, 6, 23], np.int32)
            ora.render_inline(spheres, planes, cam, w, h, 4, 1, [p[rows] for p in start(k)], rows=rows)
            for rule in (ora.SEED_FROM_RESULT, ora.SEED_KEEP_ACCUMULATOR):
                ora.render_streams(spheres, planes, cam, w, h, 1 << 10, 2, start(k), seed_rule=rule, n_threads=2)
        else:
            for cap in (1, 3, 24):
                ora.render_streams_tree(spheres, planes, cam, w, h, cap, 2, start(k), n_threads=2)
            ora.render_streams_wavefront(spheres, planes, cam, w, h, 6, 2, start(k), capacity_factor=64)
            ora.render_streams_wavefront(spheres, planes, cam, w, h, 6, 1, start(k), capacity_factor=1)        # the streams overflow: children are dropped, nothing is overrun
    a = ora.sfc32_seed3(1, 2, 3)
    ora.sfc32_stream(a, 8)
print("sanitized oracle: done")
'''


def test_the_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan here")
    lib = str(tmp_path / "libptoracle_sanitized.so")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fno-omit-frame-pointer",
           "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all", "-o", lib, os.path.join(ROOT, "oracle", "pt_oracle.c"), "-lm"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, PTMI_ORACLE_LIB=lib, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    run = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    out = run.stdout + run.stderr
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
    assert run.returncode == 0 and "sanitized oracle: done" in run.stdout, out[-3000:]
