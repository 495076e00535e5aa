"""TEST INFRASTRUCTURE ONLY: builds the CPU emulator of the generation chain (tests/emu/libwn_emu.so) from
the SAME source as the HIP library (csrc/wn_runtime.hip + wn_kernel.h + wn_plan.h) with -DWN_EMU.  It lets
the GPU-less authoring container test planner, packer, ABI argument handling and the kernel's index
arithmetic against the oracle.  The product package never loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "pytorch-wavenet_amd", "csrc")
OUT = os.path.join(HERE, "libwn_emu.so")
DEPS = [os.path.join(CSRC, f) for f in ("wn_runtime.hip", "wn_kernel.h", "wn_plan.h")] + [os.path.join(ROOT, "include", "wn_abi.h")]


def build_emu(force=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    cmd = ["g++", "-x", "c++", "-DWN_EMU", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
           "-Wno-unused-function", "-o", OUT, os.path.join(CSRC, "wn_runtime.hip")]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_emu(force=True))
