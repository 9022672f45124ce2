"""hemocell_amd -- MI355X-native IB-LBM hot path of HemoCell behind a C ABI.

csrc/  HIP kernels + the C ABI (include/hemocell_amd.h) -> lib/libhemocell_amd.so
capi   ctypes binding of that ABI
host   numpy-facing mirror of the reference's interface for this path
"""
from . import capi  # noqa: F401
