"""haskell-path-tracer_amd -- MI355X-native `render` path behind haskell-path-tracer's boundary.

Holds only what the hot path needs:
  csrc/        hand-written gfx950 HIP kernels + the C ABI of include/ptmi.h  (-> libptmi.so)
  binding.py   ctypes mirror of the C ABI (tests, bench)
  world.py     Scene.World data (mainScene, initialCamera) + the 16-primitive bench scene
  hostcxx/     C++ mirror of the reference interface (compileFor / render / initialOutput) over the C ABI
  haskell/     the foreign-import module a maintainer adds to the reference (source only)
  parallel.py  row-stripe partition over ranks + RCCL gather of the colour planes
The directory name is not a Python identifier; load it with __graft_entry__.load_package().
"""
from . import _build, world  # noqa: F401
from . import binding  # noqa: F401
from .binding import Context, Group, PtmiError, SYMBOLS, load_library  # noqa: F401
from .world import INLINE, STREAMS, MATTE, GLOSSY, GLASS  # noqa: F401
