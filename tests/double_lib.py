"""TEST INFRASTRUCTURE: the host-memory test double of the wn_abi library (tests/double/build_double.py) and the memory provider that
goes with it -- numpy arrays instead of device memory.  Host-logic tests inject both into mi355_wavenet.engine.Engine
(``Engine(..., **double_backend())``); the product package contains neither."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "double"))
_lib = None


class HostMem:
    """numpy-backed 'device' memory for the test double (it takes host pointers)."""
    def upload(self, a):
        return np.ascontiguousarray(a)

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def ptr(self, a):
        return a.ctypes.data if a is not None else None

    def download(self, a):
        return np.array(a, copy=True)

    def stream(self):
        return None


def double_library():
    global _lib
    if _lib is None:
        import build_double
        from mi355_wavenet import _abi
        _lib = _abi.Library(build_double.build_double())
    return _lib


def double_backend():
    """keyword arguments for Engine / generate_streams: the test double and its memory provider"""
    return {"lib": double_library(), "mem": HostMem()}
