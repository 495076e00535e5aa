"""TEST INFRASTRUCTURE: loads the CPU emulator build of the wn_abi library (see tests/emu/build_emu.py)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))
_lib = None


def emu_library():
    global _lib
    if _lib is None:
        import build_emu
        from mi355_wavenet import _abi
        _lib = _abi.Library(build_emu.build_emu(), host_memory=True)
    return _lib
