"""Builds libptmi.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

The flags are part of the arithmetic contract (DESIGN.md "numerics"): -ffp-contract=off keeps
every f32/f64 operation separately rounded on host and device; hipcc's default correctly
rounded sqrt/division stays on; no fast-math.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptmi.so")
ABLATIONS_LIB = os.path.join(HERE, "libptmi_ablations.so")   # the same library with the ablation kernels of DESIGN.md 5.2 (tests, measurements)
SOURCES = ["ptmi_api.cpp", "ptmi_stage.cpp", "ptmi_group.cpp", "ptmi_kernels.hip"]
HEADERS = ["ptmi_core.h", "ptmi_kernels.h", "ptmi_stage.h", os.path.join("..", "..", "include", "ptmi.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-fast-math", "-fno-slp-vectorize", "-DPTMI_SINCOS_FUSED=1", "-Wall", "-pthread", "-ldl"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libptmi.so cannot be built")


def source_hash():
    """sha256 over the kernel sources, the headers and the build flags: what a profile of the binary is a profile OF
    (profiles/*_valu_roofline.json carries it; bench.py says `stale` when the sources have moved on)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_ablations_lib(force=False):
    """libptmi_ablations.so = libptmi.so + the ablation kernels (-DPTMI_ABLATIONS): what ptmi_set_variant's other values need."""
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(ABLATIONS_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(ABLATIONS_LIB) for d in deps):
        return ABLATIONS_LIB
    import fcntl
    with open(ABLATIONS_LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        tmp = "%s.tmp.%d" % (ABLATIONS_LIB, os.getpid())
        build_lib(out=tmp, extra_flags=["-DPTMI_ABLATIONS"])
        os.replace(tmp, ABLATIONS_LIB)
    return ABLATIONS_LIB


def _contracted_object(obj, extra_flags=(), verbose=False):
    """ptmi_kernels.hip once more, render Inline only, with a * b + c contracted into FMAs (-ffp-contract=fast) and every name
    in namespace ptmi_contracted: the measurement object behind PTMI_OPT_ARITHMETIC (never the default arithmetic)."""
    flags = [f for f in FLAGS if f not in ("-ffp-contract=off", "-shared", "-pthread", "-ldl")]
    cmd = [hipcc_path()] + flags + list(extra_flags) + ["-ffp-contract=fast", "-DPTMI_CONTRACTED_BUILD", "-Dptmi=ptmi_contracted", "-c",
                                                          os.path.join(CSRC, "ptmi_kernels.hip"), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return obj


def build_lib(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the shared library if sources are newer than it. Returns its path.
    `out` + `extra_flags` build a differently-flagged copy elsewhere (diagnostic builds)."""
    if out is not None:
        obj = _contracted_object(out + ".contracted.o", extra_flags)
        cmd = [hipcc_path()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-Wl," + obj, "-o", out]
        res = subprocess.run(cmd, capture_output=True, text=True)
        os.remove(obj)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
        return out
    if not force and not is_stale():
        return LIB
    # one builder at a time: the ranks of a multi-GPU launch all come through here
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or is_stale():
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            obj = _contracted_object(tmp + ".contracted.o", extra_flags, verbose)
            cmd = [hipcc_path()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-Wl," + obj, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            os.remove(obj)
            if res.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, LIB)
    return LIB
