"""Builds the gfx950 shared library (the C ABI of include/cuhe_hip.h) in-tree.

hipcc cross-compiles without a GPU; the .so lands in cuhe_amd/lib/ so that it
travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcuhe_hip.so")
SOURCES = ["cuhe_hip.hip"]
DEPS = ["cuhe_hip.hip", "ntt_kernels.cuh", "ops_kernels.cuh", "modp.cuh", "host_math.hpp",
        os.path.join("..", "..", "include", "cuhe_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
