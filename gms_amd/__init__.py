"""gms_amd — MI355X-native set-intersection / subgraph-enumeration backend for GraphMineSuite (spcl/gms).

The product is the C-ABI shared library gms_amd/lib/libgmsx.so (include/gmsx.h): host graph substrate in
C++ (gms_amd/csrc/host) + hand-written gfx950 HIP kernels (gms_amd/csrc/hip).  `gms_amd.capi` is a thin
ctypes binding used by tests/ and bench.py.
"""
from . import capi  # noqa: F401
from .capi import DeviceGraph, GmsxError, HostCSR  # noqa: F401
