"""Builds libptmi.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

One translation unit per kernel family (csrc/ptmi_*.hip over the shared csrc/ptmi_device.h), compiled in parallel into
build/obj/<flags key>/ and linked; an edit recompiles the units whose sources or headers moved (their -MD dependency files).
The Inline unit is compiled a second time with contracted arithmetic (a labelled measurement mode, PTMI_OPT_ARITHMETIC).

The flags are part of the arithmetic contract (DESIGN.md "numerics"): -ffp-contract=off keeps
every f32/f64 operation separately rounded on host and device; hipcc's default correctly
rounded sqrt/division stays on; no fast-math.
"""
import hashlib
import mmap
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
BUILD_ID_UNIT = "ptmi_build_id.cpp"                          # ptmi_build_id(): compiled at every link with -DPTMI_BUILD_ID=<what the library was built from>
BUILD_ID_MARKER = b"PTMI_BUILD_ID="                          # ... behind this marker in the binary, so that the file can be asked without loading it
INLINE_UNIT = "ptmi_inline.hip"                              # also the contracted-arithmetic object
KERNEL_UNITS = [INLINE_UNIT, "ptmi_streams_chain.hip", "ptmi_streams_tree.hip", "ptmi_stream_primary.hip", "ptmi_stream_pixels.hip",
                "ptmi_stream_split.hip", "ptmi_small.hip"]
ABLATION_UNITS = ["ptmi_inline_ablations.hip"]               # only with -DPTMI_ABLATIONS
SOURCES = HOST_SOURCES + [BUILD_ID_UNIT] + KERNEL_UNITS + ABLATION_UNITS
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
    """sha256 over the kernel sources, the headers and the build flags: what a binary is built FROM and what a profile of
    it is a profile OF.  The library carries it (ptmi_build_id(), include/ptmi.h), binding.open_library refuses a library
    that carries another one, bench.py prints both, and profiles/*_valu_roofline.json name the one they were taken on."""
    h = hashlib.sha256(" ".join(COMPILE_FLAGS + LINK_FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def build_id(extra_flags=()):
    """What ptmi_build_id() of a library built NOW from these sources with these extra flags returns: the source hash, and
    behind a '+' the extra flags of a non-default build (ablations, diagnostic builds)."""
    extra = sorted(extra_flags)
    return source_hash() + ("+" + ",".join(f[2:] if f.startswith("-D") else f for f in extra) if extra else "")


def read_build_id(lib):
    """The id a built library carries, read from the FILE (no dlopen: a library of the same name may already be mapped), or
    None for a missing file or one without the marker (a build from before ptmi_build_id existed)."""
    try:
        with open(lib, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as m:
            at = m.find(BUILD_ID_MARKER)
            if at < 0:
                return None
            at += len(BUILD_ID_MARKER)
            end = m.find(b"\0", at, at + 512)
            return m[at:end].decode("ascii", "replace") if end > at else None
    except (OSError, ValueError):
        return None


def _all_deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS]


def is_stale(lib=LIB, extra_flags=()):
    """A library is current iff it CARRIES the id of the present sources and flags -- file times say nothing about a binary that
    travelled (the .so files ship to the GPU box with the snapshot; a checkout or a copy resets every mtime)."""
    return read_build_id(lib) != build_id(extra_flags)


def _deps_of(obj, src):
    """The files of this repository an object was compiled from (its -MD file; system and ROCm headers are covered by
    toolchain_id), or every source and header when that is missing."""
    dep = obj + ".d"
    if not os.path.exists(dep):
        return _all_deps()
    with open(dep) as fh:
        words = fh.read().replace("\\\n", " ").split()
    root = os.path.realpath(ROOT)
    files = [os.path.realpath(w) for w in words[1:] if not w.endswith(":") and os.path.exists(w)]
    return sorted(set(f for f in files if f.startswith(root + os.sep)) | {os.path.realpath(src)})


def _digest(files, flags):
    h = hashlib.sha256(" ".join(flags).encode())
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.relpath(os.path.realpath(f), os.path.realpath(ROOT)).encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()


def _compile(src, obj, flags, verbose=False, force=False):
    """One unit -> one object.  A cached object is reused when the CONTENTS of everything it was compiled from (its -MD file)
    and its flags still have the digest recorded beside it (obj.sum) -- no file times: build/obj travels to the GPU box with
    the snapshot.  `force` recompiles regardless.  The object appears under its name only when complete (two builders with
    the same flags may run side by side: tools/ab.py, the ranks of a multi-GPU launch)."""
    if not force and os.path.exists(obj) and os.path.exists(obj + ".sum"):
        with open(obj + ".sum") as fh:
            if fh.read().strip() == _digest(_deps_of(obj, src), flags):
                return obj
    tmp = "%s.tmp.%d" % (obj, os.getpid())
    cmd = [hipcc_path()] + flags + ["-MD", "-MF", tmp + ".d", "-MT", obj, "-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        for f in (tmp, tmp + ".d"):
            if os.path.exists(f):
                os.remove(f)
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if os.path.exists(obj + ".sum"):
        os.remove(obj + ".sum")                      # (never a new object beside an old digest)
    os.replace(tmp + ".d", obj + ".d")
    os.replace(tmp, obj)
    with open(tmp + ".sum", "w") as fh:
        fh.write(_digest(_deps_of(obj, src), flags) + "\n")
    os.replace(tmp + ".sum", obj + ".sum")
    return obj


_toolchain = None


def toolchain_id():
    """`hipcc --version`, hashed: objects of another compiler or ROCm release are not reused (part of the object directory's key)."""
    global _toolchain
    if _toolchain is None:
        res = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True)
        _toolchain = hashlib.sha256((res.stdout + res.stderr + os.environ.get("HIPCC_COMPILE_FLAGS_APPEND", "")).encode()).hexdigest()[:12]
    return _toolchain


def _build(out, extra_flags=(), verbose=False, force=False):
    """Every unit -> object (in parallel, cached per flag set and toolchain), then the link -- with ptmi_build_id.cpp compiled
    afresh, carrying the id of what was just compiled."""
    extra = list(extra_flags)
    ident = build_id(extra)                          # BEFORE compiling: an edit during the build makes the library stale, not wrong
    key = hashlib.sha256(" ".join(COMPILE_FLAGS + extra + [toolchain_id()]).encode()).hexdigest()[:12]
    obj_dir = os.path.join(OBJ_ROOT, key)
    os.makedirs(obj_dir, exist_ok=True)
    units = HOST_SOURCES + KERNEL_UNITS + (ABLATION_UNITS if "-DPTMI_ABLATIONS" in extra else [])
    jobs = [(os.path.join(CSRC, u), os.path.join(obj_dir, u + ".o"), COMPILE_FLAGS + extra) for u in units]
    # render Inline once more with a * b + c contracted into FMAs (-ffp-contract=fast) and every name in namespace
    # ptmi_contracted: the measurement object behind PTMI_OPT_ARITHMETIC (never the default arithmetic)
    contracted = [f for f in COMPILE_FLAGS if f != "-ffp-contract=off"] + extra + ["-ffp-contract=fast", "-DPTMI_CONTRACTED_BUILD", "-Dptmi=ptmi_contracted"]
    jobs.append((os.path.join(CSRC, INLINE_UNIT), os.path.join(obj_dir, INLINE_UNIT + ".contracted.o"), contracted))
    with ThreadPoolExecutor(JOBS) as pool:
        objs = list(pool.map(lambda j: _compile(j[0], j[1], j[2], verbose, force), jobs))
    id_obj = os.path.join(obj_dir, "%s.%d.o" % (BUILD_ID_UNIT, os.getpid()))
    host_flags = [f for f in COMPILE_FLAGS if not f.startswith("--offload-arch")]
    res = subprocess.run([hipcc_path()] + host_flags + ['-DPTMI_BUILD_ID="%s"' % ident, "-c", os.path.join(CSRC, BUILD_ID_UNIT), "-o", id_obj],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n" % BUILD_ID_UNIT + res.stdout + res.stderr)
    objs.append(id_obj)
    cmd = [hipcc_path()] + LINK_FLAGS + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    os.remove(id_obj)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    if read_build_id(out) != ident:
        raise RuntimeError("%s does not carry the build id %s it was linked with" % (out, ident))
    return out


def _locked_build(lib, extra_flags, force, verbose=False):
    if not force and not is_stale(lib, extra_flags):
        return lib
    # one builder at a time: the ranks of a multi-GPU launch all come through here
    import fcntl
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or is_stale(lib, extra_flags):
            tmp = "%s.tmp.%d" % (lib, os.getpid())
            _build(tmp, extra_flags, verbose, force)
            os.replace(tmp, lib)
    return lib


def build_ablations_lib(force=False):
    """libptmi_ablations.so = libptmi.so + the ablation kernels (-DPTMI_ABLATIONS): what ptmi_set_variant's other values need."""
    return _locked_build(ABLATIONS_LIB, ["-DPTMI_ABLATIONS"], force)


def build_lib(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the shared library unless the one in place carries the id of the present sources (`force`: recompile every
    unit regardless).  Returns its path.  `out` + `extra_flags` build a differently-flagged copy elsewhere (diagnostic builds)."""
    if out is not None:
        return _build(out, extra_flags, verbose, force)
    return _locked_build(LIB, list(extra_flags), force, verbose)
