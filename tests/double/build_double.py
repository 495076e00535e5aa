"""TEST INFRASTRUCTURE ONLY: builds tests/double/libwn_double.so, the host-memory test double of include/wn_abi.h
(wn_abi_double.cpp on top of the C oracle).  The product package never loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libwn_double.so")
SRC = os.path.join(HERE, "wn_abi_double.cpp")
ORACLE = [os.path.join(ROOT, "oracle", f) for f in ("wn_oracle.c", "wn_oracle_impl.h")]
DEPS = [SRC, os.path.join(ROOT, "include", "wn_abi.h")] + ORACLE


def build_double(force=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    obj = os.path.join(HERE, "wn_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c99", "-ffp-contract=off", "-Wall", "-c", "-o", obj, ORACLE[0]])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", OUT, SRC, obj, "-lm"])
    os.remove(obj)
    return OUT


if __name__ == "__main__":
    print(build_double(force=True))
