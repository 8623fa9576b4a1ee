"""Builds libptmi.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

One translation unit per kernel family (csrc/ptmi_*.hip over the shared csrc/ptmi_device.h), compiled in parallel into
build/obj/<flags key>/ and linked; an edit recompiles the units whose sources or headers moved (their -MD dependency files).
The Inline unit is compiled a second time with contracted arithmetic (a labelled measurement mode, PTMI_OPT_ARITHMETIC).

The flags are part of the arithmetic contract (DESIGN.md "numerics"): -ffp-contract=off keeps
every f32/f64 operation separately rounded on host and device; hipcc's default correctly
rounded sqrt/division stays on; no fast-math.
"""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptmi.so")
ABLATIONS_LIB = os.path.join(HERE, "libptmi_ablations.so")   # the same library with the ablation kernels of DESIGN.md 5.2 (tests, measurements)
OBJ_ROOT = os.path.join(ROOT, "build", "obj")
HOST_SOURCES = ["ptmi_api.cpp", "ptmi_stage.cpp", "ptmi_group.cpp"]
INLINE_UNIT = "ptmi_inline.hip"                              # also the contracted-arithmetic object
KERNEL_UNITS = [INLINE_UNIT, "ptmi_streams_chain.hip", "ptmi_streams_tree.hip", "ptmi_stream_primary.hip", "ptmi_stream_pixels.hip",
                "ptmi_stream_split.hip", "ptmi_small.hip"]
ABLATION_UNITS = ["ptmi_inline_ablations.hip"]               # only with -DPTMI_ABLATIONS
SOURCES = HOST_SOURCES + KERNEL_UNITS + ABLATION_UNITS
HEADERS = ["ptmi_core.h", "ptmi_kernels.h", "ptmi_device.h", "ptmi_diag.h", "ptmi_stream_form.h", "ptmi_stage.h",
           os.path.join("..", "..", "include", "ptmi.h")]
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                 "-fno-fast-math", "-fno-slp-vectorize", "-DPTMI_SINCOS_FUSED=1", "-Wall", "-pthread"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-ldl"]
FLAGS = COMPILE_FLAGS + ["-shared", "-ldl"]                  # (what tools that compile a single unit start from)
JOBS = max(1, min(8, os.cpu_count() or 1))


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libptmi.so cannot be built")


def source_hash():
    """sha256 over the kernel sources, the headers and the build flags: what a profile of the binary is a profile OF
    (profiles/*_valu_roofline.json carries it; bench.py says `stale` when the sources have moved on)."""
    h = hashlib.sha256(" ".join(COMPILE_FLAGS + LINK_FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _all_deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]


def is_stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _all_deps())


def _deps_of(obj, src):
    """The files an object was compiled from (its -MD file), or every source and header when that is missing."""
    dep = obj + ".d"
    if not os.path.exists(dep):
        return _all_deps()
    with open(dep) as fh:
        words = fh.read().replace("\\\n", " ").split()
    files = [w for w in words[1:] if not w.endswith(":") and os.path.exists(w)]
    return files + [src, os.path.abspath(__file__)]


def _compile(src, obj, flags, verbose=False):
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps_of(obj, src)):
        return obj
    cmd = [hipcc_path()] + flags + ["-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    return obj


def _build(out, extra_flags=(), verbose=False):
    """Every unit -> object (in parallel, cached per flag set), then the link."""
    extra = list(extra_flags)
    key = hashlib.sha256(" ".join(COMPILE_FLAGS + extra).encode()).hexdigest()[:12]
    obj_dir = os.path.join(OBJ_ROOT, key)
    os.makedirs(obj_dir, exist_ok=True)
    units = HOST_SOURCES + KERNEL_UNITS + (ABLATION_UNITS if "-DPTMI_ABLATIONS" in extra else [])
    jobs = [(os.path.join(CSRC, u), os.path.join(obj_dir, u + ".o"), COMPILE_FLAGS + extra) for u in units]
    # render Inline once more with a * b + c contracted into FMAs (-ffp-contract=fast) and every name in namespace
    # ptmi_contracted: the measurement object behind PTMI_OPT_ARITHMETIC (never the default arithmetic)
    contracted = [f for f in COMPILE_FLAGS if f != "-ffp-contract=off"] + extra + ["-ffp-contract=fast", "-DPTMI_CONTRACTED_BUILD", "-Dptmi=ptmi_contracted"]
    jobs.append((os.path.join(CSRC, INLINE_UNIT), os.path.join(obj_dir, INLINE_UNIT + ".contracted.o"), contracted))
    with ThreadPoolExecutor(JOBS) as pool:
        objs = list(pool.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    cmd = [hipcc_path()] + LINK_FLAGS + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    return out


def _locked_build(lib, extra_flags, force, verbose=False):
    if not force and not is_stale(lib):
        return lib
    # one builder at a time: the ranks of a multi-GPU launch all come through here
    import fcntl
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or is_stale(lib):
            tmp = "%s.tmp.%d" % (lib, os.getpid())
            _build(tmp, extra_flags, verbose)
            os.replace(tmp, lib)
    return lib


def build_ablations_lib(force=False):
    """libptmi_ablations.so = libptmi.so + the ablation kernels (-DPTMI_ABLATIONS): what ptmi_set_variant's other values need."""
    return _locked_build(ABLATIONS_LIB, ["-DPTMI_ABLATIONS"], force)


def build_lib(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the shared library if sources are newer than it. Returns its path.
    `out` + `extra_flags` build a differently-flagged copy elsewhere (diagnostic builds)."""
    if out is not None:
        return _build(out, extra_flags, verbose)
    return _locked_build(LIB, list(extra_flags), force, verbose)
