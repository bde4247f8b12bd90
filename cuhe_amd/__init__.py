"""cuhe_amd -- MI355X-native large-polynomial arithmetic backend behind cuHE's API.

The product is the C-ABI shared library (include/cuhe_hip.h, built from
cuhe_amd/csrc/) plus the C++ drop-in headers in cuhe_amd/cxx/.  This Python
package only exposes a ctypes binding of that ABI for tests and bench.py; there
is no CPU fallback: importing `cuhe_amd.capi` without the built library raises.
"""
__all__ = ["capi", "build"]
