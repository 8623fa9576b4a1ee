"""The HOST side of libptmi under AddressSanitizer + UndefinedBehaviorSanitizer, here, without a GPU (GPU sanitizers are not available on the
pool; CPU ones are).  libptmi is built once more with its host code instrumented (-Xarch_host -fsanitize=address,undefined; the gfx950 code
is the same) and driven, in a CHILD process, on a test-only stand-in for the HIP runtime (tests/cxx/hip_stub.cpp: device memory is host
memory, copies are memcpy, kernels do not run) and the RCCL stand-in of tests/test_gpu_group_rccl_stub.py.  tests/hostsan_driver.py takes it
through the resident path, partitions, GLASS and the stream form's bookkeeping (with the counts its kernels would have read back PLACED
into the read-backs: overflow levels, the redo with longer streams, drops that stand), both closures with evictions / consumption / stale tokens,
staged copies through the pinned ring's worker threads, four threads on one context, groups of 1-8 members with their gathers, and the
refusals -- first plainly, then once per FAILURE POINT: the k-th hipMalloc, copy, launch, synchronize, pinned allocation or stream/event
creation of the scenario fails, for every k; then the k-th AND the next call of the kind (the recovery's own call fails), and the k-th and
every later one (a device that stays broken), for every fifth k (about 3 500 runs, ~25 s).  Demanded of every run: no sanitizer report (an overrun copy, a block used after hipFree or freed twice, a
wild stream handle, signed overflow, ...); after the contexts are destroyed the stand-in holds no device block, pinned block, stream or
event; no error the library reported or chose to ignore is left in the runtime's sticky slot (hipGetLastError), and no launch of the library
is ever blamed for an older call's error.

Then a seeded RANDOM WALK over the ABI (tests/hostsan_driver.py: monkey) that keeps using the context and the group after every failure.
What IS checked for value: everything that is copies only (`roundtrips`: uploads, downloads, the closures with kernels that do nothing,
evictions to host memory and back, snapshots, the group's stitched read-out) -- also with the stand-in's streams asynchronous.

What this does NOT test: any rendered value, any kernel, the HIP runtime or RCCL themselves.  The product never meets the stand-in: it is
preloaded into the child process only, and libptmi has no CPU path (tests/test_abi.py)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402

OUT = os.path.join(ROOT, "build", "hip_stub")
STUB = os.path.join(OUT, "libhipstub.so")
STUB_SRC = os.path.join(ROOT, "tests", "cxx", "hip_stub.cpp")
SANITIZED = os.path.join(OUT, "libptmi_sanitized.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HOST_SANITIZE = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-shared-libasan"]


def runtime(which):
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.%s-x86_64.so" % which))
    return found[-1] if found else None


def asan_runtime():
    return runtime("asan")


def build_stub(out=STUB, sanitize="address,undefined"):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(STUB_SRC):
        return out
    cmd = [CLANG, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Werror",
           "-fsanitize=" + sanitize, "-shared-libsan", "-fno-omit-frame-pointer", STUB_SRC, "-o", out + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    os.replace(out + ".tmp", out)
    return out


def dynamic_symbols(lib, undefined):
    out = subprocess.run(["nm", "-D", "--undefined-only" if undefined else "--defined-only", lib], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return {line.split()[-1].split("@")[0] for line in out.stdout.splitlines() if line.strip()}


needs_asan = pytest.mark.skipif(asan_runtime() is None or not os.path.exists(CLANG), reason="the ROCm clang has no x86-64 ASan runtime here")


@needs_asan
def test_the_stand_in_covers_every_runtime_call_libptmi_makes():
    """No call of the library may fall through to the real runtime in the child process: every hip* / __hip* symbol libptmi.so imports is
    one the stand-in defines (so a new runtime call in csrc/ fails HERE until the stand-in knows it)."""
    pkg = graft.load_package()
    lib = pkg._build.build_lib()
    wanted = {s for s in dynamic_symbols(lib, True) if s.startswith("hip") or s.startswith("__hip")}
    assert len(wanted) >= 30, wanted
    missing = wanted - dynamic_symbols(build_stub(), False)
    assert not missing, "tests/cxx/hip_stub.cpp lacks %s" % sorted(missing)
    assert not any("nccl" in s.lower() for s in dynamic_symbols(lib, True))      # RCCL is resolved with dlopen: the RCCL stand-in serves it


@needs_asan
def test_host_side_is_clean_under_asan_and_ubsan_at_every_failure_point():
    import test_gpu_group_rccl_stub as rccl
    pkg = graft.load_package()
    stub = build_stub()
    rccl_dir = os.path.dirname(rccl.build_stub())
    lib = pkg._build.build_lib(out=SANITIZED, extra_flags=HOST_SANITIZE)
    assert "__asan_init" in dynamic_symbols(lib, True)              # the host code really is instrumented
    env = dict(os.environ, PTMI_HIPSTUB=stub, PTMI_SANITIZED_LIB=lib, LD_PRELOAD="%s %s" % (asan_runtime(), stub),
               LD_LIBRARY_PATH=rccl_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("PTMI_HOSTSAN_ONLY", None)
    env.pop("PTMI_HOSTSAN_STRIDE", None)
    env["PTMI_HOSTSAN_MORE_STRIDE"] = "5"                          # (every k alone; every fifth k with the next / with all later calls failing too)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_driver.py")], capture_output=True, text=True, env=env, timeout=900)
    out = run.stdout + run.stderr
    assert "runtime error" not in out and "AddressSanitizer" not in out and "HIPSTUB:" not in out, out[-4000:]
    assert run.returncode == 0 and "sanitized host side: done" in run.stdout, out[-4000:]
    reports = [line for line in run.stdout.splitlines() if line.startswith("hostsan ")]
    assert len(reports) == 10, reports
    walked = sum(int(line.split("'failure_points_walked': ")[1].split(",")[0]) for line in reports)
    assert walked >= 3000, walked
    print("\n".join(reports))
    # ... and a seeded random walk over the whole ABI (any order of calls, any arguments, an injected failure every ~30 steps, the context
    # USED ON after every failure -- which the walk above, whose scenarios end at their first error, never does): 60 seeds x 1 500 steps
    env["PTMI_HOSTSAN_ONLY"] = "monkey:0:60:1500"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_driver.py")], capture_output=True, text=True, env=env, timeout=900)
    out = run.stdout + run.stderr
    assert "runtime error" not in out and "AddressSanitizer" not in out and "HIPSTUB:" not in out and "terminate called" not in out, out[-4000:]
    assert run.returncode == 0 and "sanitized host side: done" in run.stdout and out.count("hostsan monkey seed") == 60, out[-4000:]
    # ... and all of it once more with the stand-in's streams truly ASYNCHRONOUS (hipstub_set_deferred: a pinned copy, a device-to-device copy, a
    # fill happen at the next synchronisation that covers them): the value-checked round trips of the `roundtrips` scenario then hold only if the
    # library synchronises before it reads a result, reuses a slot of its pinned ring or returns a borrowed buffer
    env["PTMI_HOSTSAN_DEFERRED"] = "1"
    for only, stride, want in ((None, "4", 10), ("monkey:200:30:1500", "1", 30)):
        env.pop("PTMI_HOSTSAN_ONLY", None)
        if only:
            env["PTMI_HOSTSAN_ONLY"] = only
        env["PTMI_HOSTSAN_STRIDE"] = stride
        run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_driver.py")], capture_output=True, text=True, env=env, timeout=900)
        out = run.stdout + run.stderr
        assert "runtime error" not in out and "AddressSanitizer" not in out and "HIPSTUB:" not in out and "terminate called" not in out, out[-4000:]
        assert run.returncode == 0 and "sanitized host side: done" in run.stdout and len([x for x in run.stdout.splitlines() if x.startswith("hostsan ")]) == want, out[-4000:]
    env.pop("PTMI_HOSTSAN_DEFERRED")
    env.pop("PTMI_HOSTSAN_STRIDE")
    # ... and from three threads at once on one context (each thread must also be handed ITS OWN failure's message)
    env["PTMI_HOSTSAN_ONLY"] = "monkey_threads:0:10:300"
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_driver.py")], capture_output=True, text=True, env=env, timeout=900)
    out = run.stdout + run.stderr
    assert "runtime error" not in out and "AddressSanitizer" not in out and "HIPSTUB:" not in out and "terminate called" not in out, out[-4000:]
    assert run.returncode == 0 and "sanitized host side: done" in run.stdout and out.count("hostsan monkey seed") == 10, out[-4000:]


@pytest.mark.skipif(runtime("tsan") is None or not os.path.exists(CLANG), reason="the ROCm clang has no x86-64 TSan runtime here")
def test_host_threads_are_clean_under_tsan():
    """The same, with the host code under ThreadSanitizer, on the scenarios in which host threads meet: four application threads on one
    context through the chained closure (calls, fetches, releases, resident renders), a group's per-member threads, the pinned ring's
    workers.  (The setup does report a race when there is one: checked by hand with a deliberately racy library under the same preload.)"""
    import test_gpu_group_rccl_stub as rccl
    pkg = graft.load_package()
    out = os.path.join(ROOT, "build", "hip_stub_tsan")
    stub = build_stub(os.path.join(out, "libhipstub.so"), "thread")
    rccl_dir = os.path.dirname(rccl.build_stub())
    lib = pkg._build.build_lib(out=os.path.join(out, "libptmi_tsan.so"), extra_flags=["-Xarch_host", "-fsanitize=thread", "-fno-omit-frame-pointer", "-g", "-shared-libsan"])
    assert "__tsan_init" in dynamic_symbols(lib, True)
    env = dict(os.environ, PTMI_HIPSTUB=stub, PTMI_SANITIZED_LIB=lib, LD_PRELOAD="%s %s" % (runtime("tsan"), stub),
               LD_LIBRARY_PATH=rccl_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""),
               TSAN_OPTIONS="halt_on_error=0:exitcode=24:report_signal_unsafe=0", PTMI_HOSTSAN_ONLY="threads,staged,group", PTMI_HOSTSAN_STRIDE="1000000")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_driver.py")], capture_output=True, text=True, env=env, timeout=900)
    out_text = run.stdout + run.stderr
    assert "ThreadSanitizer" not in out_text and "HIPSTUB:" not in out_text, out_text[-4000:]
    assert run.returncode == 0 and "sanitized host side: done" in run.stdout, out_text[-4000:]
    env["PTMI_HOSTSAN_ONLY"] = "monkey_threads:0:10:300"            # the random walk from three threads on one context
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsan_driver.py")], capture_output=True, text=True, env=env, timeout=900)
    out_text = run.stdout + run.stderr
    assert "ThreadSanitizer" not in out_text and "HIPSTUB:" not in out_text, out_text[-4000:]
    assert run.returncode == 0 and "sanitized host side: done" in run.stdout, out_text[-4000:]


@needs_asan
def test_the_cxx_host_mirror_is_clean_under_asan_and_leak_free(ora):
    """hostcxx/scene.hpp (the C++ mirror of the reference's closure: lazy RenderResults that hold tokens, released by destructors) through the
    whole flow of tests/cxx/host_mirror_test.cpp -- compileFor's 100 chained calls, reseed, camera move, a RenderResult from the host, the
    copying closure, the resident flow -- as a NATIVE program under ASan + UBSan + LeakSanitizer (a native program can have leak detection:
    "device" blocks are heap blocks on the stand-in, so a state or plane never given back is a reported leak).  Its comparisons with the
    oracle FAIL here by construction (kernels do not run on the stand-in): the GPU test of tests/test_host_cxx.py checks the values."""
    pkg = graft.load_package()
    stub = build_stub()
    lib = pkg._build.build_lib(out=SANITIZED, extra_flags=HOST_SANITIZE)
    exe = os.path.join(OUT, "host_mirror_sanitized")
    rt_dir = os.path.dirname(asan_runtime())
    cmd = [CLANG, "-std=c++17", "-O1", "-g", "-Wall", "-fsanitize=address,undefined", "-shared-libsan", "-fno-omit-frame-pointer",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "haskell-path-tracer_amd", "hostcxx"),
           os.path.join(ROOT, "tests", "cxx", "host_mirror_test.cpp"), "-o", exe, lib, ora.LIB,
           "-Wl,-rpath," + OUT, "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + rt_dir, "-Wl,--allow-shlib-undefined", "-fopenmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    supp = os.path.join(OUT, "lsan.supp")
    with open(supp, "w") as fh:
        fh.write("leak:libomp\nleak:libgomp\n")                    # (the OpenMP runtime's own, under the oracle)
    env = dict(os.environ, LD_PRELOAD="%s %s" % (asan_runtime(), stub), ASAN_OPTIONS="detect_leaks=1:exitcode=23",
               LSAN_OPTIONS="suppressions=%s:print_suppressions=0" % supp, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    out = run.stdout + run.stderr
    assert "Sanitizer" not in out and "runtime error" not in out and "HIPSTUB:" not in out, out[-4000:]
    assert run.returncode == 1 and "host mirror FAILED" in out, out[-2000:]       # the flow ran to its end; values are the GPU test's business
