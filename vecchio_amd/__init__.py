"""vecchio_amd — MI355X-native per-pixel sample loop of browserdotsys/vecchio.

The product is native: vecchio_amd/csrc (HIP megakernel + C ABI, include/vecchio_amd.h) and
vecchio_amd/host (C++ mirror of the reference's Rust host side).  This package is only the
ctypes plumbing that tests and bench.py use to reach them.
"""
from . import ffi  # noqa: F401
from .scene import DeviceScene, HostScene  # noqa: F401
