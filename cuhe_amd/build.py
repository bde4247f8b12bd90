"""Builds the gfx950 shared library (the C ABI of include/cuhe_hip.h) in-tree.

hipcc cross-compiles without a GPU; the .so lands in cuhe_amd/lib/ so that it
travels to the GPU box with the repo snapshot.  The device code is first compiled
to assembly and run through tools/asm_hazard_check.py (VALU-writes-SGPR ->
VALU-reads hazards behind inline asm, which nothing pads automatically): a build
with findings is refused, so an unsafe variant of the field arithmetic cannot ship."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcuhe_hip.so")
SOURCES = ["cuhe_hip.hip"]
DEPS = ["cuhe_hip.hip", "ntt_kernels.cuh", "ops_kernels.cuh", "modp.cuh", "host_math.hpp", "comm.hpp",
        os.path.join("..", "..", "include", "cuhe_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value"]
LINK = ["-fPIC", "-shared"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(os.path.join(CSRC, d)) and os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    asm = os.path.join(LIBDIR, "device_gfx950.s")
    cmd = [HIPCC] + FLAGS + ["--cuda-device-only", "-S", "-o", asm] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_check.py"), "--asm", asm], capture_output=True, text=True)
    if verbose or chk.returncode:
        print(chk.stdout.strip()[-3000:], flush=True)
    if chk.returncode:
        raise RuntimeError("inline-asm hazard check failed: the library is not built")
    cmd = [HIPCC] + FLAGS + LINK + srcs + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
