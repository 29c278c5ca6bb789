"""eggshell_amd -- MI355X-native constraint solve for teenylasers/eggshell.

The product is the C-ABI library (include/eggshell_amd.h, built from
eggshell_amd/csrc into eggshell_amd/libeggshell_amd.so) plus the C++ adapter in
eggshell_amd/host that keeps the reference's Ensemble/Body/Constraint API.
This Python package is plumbing for tests and benchmarks: a ctypes binding
(capi) and synthetic scene generators (scenes).
"""
from . import capi, scenes  # noqa: F401
