"""ace-compiler_amd: MI355X-native RNS-CKKS runtime for ACE-generated FHE programs.

The product is the C-ABI shared library built from csrc/ (see include/acehip.h and, for the
source-level drop-in, include/rt_ant/).  This Python package is only the build driver and a thin
ctypes binding used by the tests and bench.py; it contains no arithmetic and has no CPU fallback.
"""
from .build import build, LIB  # noqa: F401
from .binding import AceHip, AceHipError, load_library  # noqa: F401
