"""vvenc_amd — MI355X (gfx950) back-end for VVenC's block-level RDO hot path.

The product is the C-ABI shared library `libvvenc_hip.so` (include/vvenc_hip.h, sources in
vvenc_amd/csrc/).  This package is the thin Python host layer used by the tests and by bench.py:
ctypes bindings (`vvenc_amd.lib`) plus `vvenc_amd.hotpath.HotPath`, which keeps pictures, work
lists and results in HBM as torch tensors and forwards raw device pointers to the library.

There is no CPU fallback: importing works anywhere (so the CPU test tier can check that the
library loads and exports its symbols), but creating a `HotPath` without a GPU raises.
"""
from .lib import load_library, LIB_PATH, VVHipError  # noqa: F401

__all__ = ["load_library", "LIB_PATH", "VVHipError"]
