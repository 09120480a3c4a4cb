"""libcimbar_amd -- MI355X (gfx950) frame-decode path for libcimbar mode B.

The product is the C-ABI shared library (include/cimbar_hip.h, libcimbar_amd/csrc/cimbar_hip.hip) and the C++ host
adapter in libcimbar_amd/host/. This Python package is plumbing: a ctypes binding used by tests, bench.py and the
multi-GPU driver, plus the synthetic-frame generator. There is no CPU decode path in here.
"""
from .decoder import HipDecoder, CimbarHipError, load_library  # noqa: F401
from . import modeb  # noqa: F401
