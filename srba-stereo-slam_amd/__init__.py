"""MI355X-native srba-stereo-slam hot path: ctypes view of the C-ABI (include/ssk_capi.h).

The product is libssk_hip.so (HIP kernels for gfx950 + the extern "C" boundary) and the C++ facade in
host/.  This package only loads the library for tests, bench.py and __graft_entry__; it contains no
numerical code and no CPU fallback: if the library is missing or no gfx950 device is present, loading
or Context() raises.
"""
from . import dist  # noqa: F401
from .capi import (  # noqa: F401
    KEYPOINT_DTYPE, DMATCH_DTYPE, SskError, Context, lib_path, load_library, declared_symbols,
)
