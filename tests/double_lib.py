"""TEST INFRASTRUCTURE: loads the host-memory test double of the wn_abi library (tests/double/build_double.py)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "double"))
_lib = None


def double_library():
    global _lib
    if _lib is None:
        import build_double
        from mi355_wavenet import _abi
        _lib = _abi.Library(build_double.build_double(), host_memory=True)
    return _lib
