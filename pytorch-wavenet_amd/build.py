"""Builds libwn_mi355.so (the HIP/gfx950 engine behind include/wn_abi.h) in-tree.

    python pytorch-wavenet_amd/build.py            # build if stale
    python pytorch-wavenet_amd/build.py --force

hipcc cross-compiles gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  gfx950 only: no other --offload-arch, no fallback.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "mi355_wavenet", "libwn_mi355.so")
SOURCES = [os.path.join(CSRC, "wn_runtime.hip")]
DEPS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inl"))) + [os.path.join(ROOT, "include", "wn_abi.h")]


def _stale(out, deps):
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build the gfx950 engine)")


def build_hip(force=False, verbose=False, extra_flags=()):
    if not force and not _stale(OUT, DEPS):
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", *extra_flags, "-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
