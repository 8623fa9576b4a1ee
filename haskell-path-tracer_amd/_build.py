"""Builds libptmi.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.

The library names itself by its CODE (ptmi_build_id() = code_id(): a hash over the compiled objects' allocated sections), not by
the text it was compiled from: profiles and fuzz records stay valid across comment and documentation edits.

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
BUILD_ID_UNIT = "ptmi_build_id.cpp"                          # ptmi_build_id(): compiled at every link with -DPTMI_BUILD_ID=<the code the library holds>
BUILD_ID_MARKER = b"PTMI_BUILD_ID="                          # ... behind this marker in the binary, so that the file can be asked without loading it
SOURCE_HASH_MARKER = b"PTMI_SOURCE_HASH="                    # ... and the hash of the source TEXT it was linked from (staleness only, see source_hash)
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
RECIPE = "cwd=csrc,cuid=unit+flags"                          # how _compile invokes the compiler, as far as the object depends on it
JOBS = max(1, min(8, os.cpu_count() or 1))


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libptmi.so cannot be built")


def source_hash():
    """sha256 over the TEXT of the kernel sources, the headers and the build flags.  It answers one question only -- "has any
    byte moved since this library was linked?" (is_stale's fast path) -- and names nothing: a comment edit changes it.  What a
    binary IS, and what a profile of it is a profile OF, is code_id() below."""
    h = hashlib.sha256(" ".join(COMPILE_FLAGS + LINK_FLAGS + [RECIPE]).encode())
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def object_code_hash(obj):
    """sha256 over the ALLOCATED sections of one ELF object (name, then contents): host .text / .rodata / .data, and .hip_fatbin --
    the gfx950 code objects.  Symbol tables, string tables, relocations and .comment are left out, so an object says the same
    after a comment edit, a re-flowed header or a renamed static helper; it says something else as soon as an instruction, a
    constant or a kernel's name (the fatbin carries them; profiles are keyed by them) moves.  Units are compiled from within
    csrc/ under their bare names (_compile), so no path of the checkout is in the object either."""
    with open(obj, "rb") as fh:
        data = fh.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise RuntimeError("%s is not a little-endian ELF64 object" % obj)
    shoff = int.from_bytes(data[0x28:0x30], "little")
    shentsize, shnum, shstrndx = (int.from_bytes(data[o:o + 2], "little") for o in (0x3A, 0x3C, 0x3E))

    def header(i):
        b = data[shoff + i * shentsize: shoff + (i + 1) * shentsize]
        return (int.from_bytes(b[0:4], "little"), int.from_bytes(b[4:8], "little"), int.from_bytes(b[8:16], "little"),
                int.from_bytes(b[24:32], "little"), int.from_bytes(b[32:40], "little"))     # name, type, flags, offset, size
    _, _, _, str_off, str_size = header(shstrndx)
    names = data[str_off:str_off + str_size]
    h = hashlib.sha256()
    for i in range(shnum):
        name, typ, flags, off, size = header(i)
        if not flags & 2 or typ == 8:                       # SHF_ALLOC only; SHT_NOBITS has no bytes
            continue
        h.update(names[name:names.index(b"\0", name)] + b"\0" + size.to_bytes(8, "little") + data[off:off + size])
    return h.hexdigest()


def code_id_of(objs):
    """The id of a library linked from these objects: sha256 over (unit name, object_code_hash) in name order, 16 hex digits."""
    h = hashlib.sha256()
    for obj in sorted(objs, key=os.path.basename):
        h.update(os.path.basename(obj).encode() + b"\0" + object_code_hash(obj).encode() + b"\0")
    return h.hexdigest()[:16]


def _flag_suffix(extra_flags=()):
    extra = sorted(extra_flags)
    return "+" + ",".join(f[2:] if f.startswith("-D") else f for f in extra) if extra else ""


def code_id(extra_flags=(), verbose=False):
    """What ptmi_build_id() of a library built NOW from these sources with these extra flags returns: the hash of the COMPILED
    code (object_code_hash of every unit; units whose sources moved are recompiled first, the others come from the object
    cache), and behind a '+' the extra flags of a non-default build (ablations, diagnostic builds).  Profiles, fuzz records and
    soaks name this id: it survives comment, documentation and rename-only edits, and nothing else."""
    return code_id_of(_compile_units(list(extra_flags), verbose, False)) + _flag_suffix(extra_flags)


def build_id(extra_flags=()):
    """code_id under its old name (what ptmi_build_id() of a fresh build returns)."""
    return code_id(extra_flags)


def _read_marked(lib, marker):
    try:
        with open(lib, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as m:
            at = m.find(marker)
            if at < 0:
                return None
            at += len(marker)
            end = m.find(b"\0", at, at + 512)
            return m[at:end].decode("ascii", "replace") if end > at else None
    except (OSError, ValueError):
        return None


def read_build_id(lib):
    """The id a built library carries, read from the FILE (no dlopen: a library of the same name may already be mapped), or
    None for a missing file or one without the marker (a build from before ptmi_build_id existed)."""
    return _read_marked(lib, BUILD_ID_MARKER)


def read_source_hash(lib):
    """source_hash() of the text a built library was linked from (read from the file), or None."""
    return _read_marked(lib, SOURCE_HASH_MARKER)


def _all_deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS]


def is_stale(lib=LIB, extra_flags=()):
    """A library is current iff it was linked from the present source text and flags (the fast path: no compiler runs), or -- the
    text having moved -- it still CARRIES the id of the code the present sources compile to (a comment was edited: code_id
    recompiles the touched units, nothing is relinked).  File times say nothing about a binary that travelled (the .so files
    ship to the GPU box with the snapshot; a checkout or a copy resets every mtime)."""
    carried = read_build_id(lib)
    if carried is None or not carried.endswith(_flag_suffix(extra_flags)) or ("+" in carried) != bool(list(extra_flags)):
        return True
    if read_source_hash(lib) == source_hash():
        return False
    return carried != code_id(extra_flags)


def matches_sources(lib):
    """True iff the library at `lib` holds the code the sources beside this file compile to, under whatever extra flags its id
    names: binding.open_library's question."""
    carried = read_build_id(lib)
    if carried is None:
        return False
    if read_source_hash(lib) == source_hash():
        return True
    extra = [f if f.startswith("-") else "-D" + f for f in carried.split("+", 1)[1].split(",")] if "+" in carried else []
    return carried == code_id(extra)


def _deps_of(obj, src):
    """The files of this repository an object was compiled from (its -MD file; system and ROCm headers are covered by
    toolchain_id), or every source and header when that is missing."""
    dep = obj + ".d"
    if not os.path.exists(dep):
        return _all_deps()
    with open(dep) as fh:
        words = fh.read().replace("\\\n", " ").split()
    root = os.path.realpath(ROOT)
    words = [os.path.join(CSRC, w) for w in words[1:] if not w.endswith(":")]      # (the units are compiled from within csrc/)
    files = [os.path.realpath(w) for w in words if os.path.exists(w)]
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
    # compiled from WITHIN csrc/ under the unit's bare name: the device code object records the name the compiler was given, and
    # an object must not depend on where the checkout lies (object_code_hash)
    # ... nor on the name of its temporary output: clang derives the compilation unit's id (a symbol's name, host and device side) from
    # its command line unless it is given one -- the unit's name and flags, hashed
    cuid = hashlib.sha256((os.path.basename(obj) + " " + " ".join(flags)).encode()).hexdigest()[:16]
    cmd = [hipcc_path()] + flags + ["-cuid=" + cuid, "-MD", "-MF", tmp + ".d", "-MT", obj, "-c", os.path.relpath(src, CSRC), "-o", tmp]
    if verbose:
        print("(cd %s && %s)" % (CSRC, " ".join(cmd)), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
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
    """Which compiler made the cached objects: objects of another compiler or ROCm release are not reused (part of the object directory's
    key).  Read from FILES -- the ROCm release's version file, the resolved name of the clang binary (it carries the version), the bytes of
    the hipcc driver -- and never by running `hipcc --version`: code_id() is asked by bench.py while a profiler's preloaded library has
    already initialised the GPU in the process, and starting another program from there is what the GPU boxes refuse."""
    global _toolchain
    if _toolchain is None:
        h = hashlib.sha256(os.environ.get("HIPCC_COMPILE_FLAGS_APPEND", "").encode())
        hipcc = os.path.realpath(hipcc_path())
        root = os.path.dirname(os.path.dirname(hipcc))
        for info in (os.path.join(root, ".info", "version"), os.path.join(root, ".info", "version-dev")):
            if os.path.exists(info):
                with open(info, "rb") as fh:
                    h.update(fh.read())
        for clang in (os.path.join(root, "lib", "llvm", "bin", "clang++"), os.path.join(root, "llvm", "bin", "clang++")):
            if os.path.exists(clang):
                h.update(os.path.realpath(clang).encode() + str(os.path.getsize(clang)).encode())
        with open(hipcc, "rb") as fh:
            h.update(fh.read())
        _toolchain = h.hexdigest()[:12]
    return _toolchain


def _compile_units(extra, verbose=False, force=False):
    """Every unit of a build with these extra flags -> its object (in parallel, cached per flag set and toolchain; a cached
    object is reused while the contents of everything it was compiled from are unchanged)."""
    key = hashlib.sha256(" ".join(COMPILE_FLAGS + extra + [toolchain_id(), RECIPE]).encode()).hexdigest()[:12]
    obj_dir = os.path.join(OBJ_ROOT, key)
    os.makedirs(obj_dir, exist_ok=True)
    units = HOST_SOURCES + KERNEL_UNITS + (ABLATION_UNITS if "-DPTMI_ABLATIONS" in extra else [])
    jobs = [(os.path.join(CSRC, u), os.path.join(obj_dir, u + ".o"), COMPILE_FLAGS + extra) for u in units]
    # render Inline once more with a * b + c contracted into FMAs (-ffp-contract=fast) and every name in namespace
    # ptmi_contracted: the measurement object behind PTMI_OPT_ARITHMETIC (never the default arithmetic)
    contracted = [f for f in COMPILE_FLAGS if f != "-ffp-contract=off"] + extra + ["-ffp-contract=fast", "-DPTMI_CONTRACTED_BUILD", "-Dptmi=ptmi_contracted"]
    jobs.append((os.path.join(CSRC, INLINE_UNIT), os.path.join(obj_dir, INLINE_UNIT + ".contracted.o"), contracted))
    with ThreadPoolExecutor(JOBS) as pool:
        return list(pool.map(lambda j: _compile(j[0], j[1], j[2], verbose, force), jobs))


def _build(out, extra_flags=(), verbose=False, force=False):
    """Every unit -> object, then the link -- with ptmi_build_id.cpp compiled afresh, carrying the id of the objects' code
    (code_id_of) and the hash of the source text they were compiled from."""
    extra = list(extra_flags)
    text = source_hash()                             # BEFORE compiling: an edit during the build makes the library stale, not wrong
    objs = _compile_units(extra, verbose, force)
    ident = code_id_of(objs) + _flag_suffix(extra)
    id_obj = os.path.join(os.path.dirname(objs[0]), "%s.%d.o" % (BUILD_ID_UNIT, os.getpid()))
    host_flags = [f for f in COMPILE_FLAGS if not f.startswith("--offload-arch")]
    res = subprocess.run([hipcc_path()] + host_flags + ['-DPTMI_BUILD_ID="%s"' % ident, '-DPTMI_SOURCE_HASH="%s"' % text,
                                                       "-c", os.path.join(CSRC, BUILD_ID_UNIT), "-o", id_obj],
                         capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n" % BUILD_ID_UNIT + res.stdout + res.stderr)
    cmd = [hipcc_path()] + LINK_FLAGS + objs + [id_obj, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    os.remove(id_obj)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    if read_build_id(out) != ident or read_source_hash(out) != text:
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
