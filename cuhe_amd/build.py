"""Builds the gfx950 shared library (the C ABI of include/cuhe_hip.h) in-tree.

hipcc cross-compiles without a GPU; the .so lands in cuhe_amd/lib/ so that it
travels to the GPU box with the repo snapshot.  The translation units (the C ABI +
two-pass kernels; the one-workgroup transforms, one unit per sub-transform size)
are compiled in parallel.  Each unit's device code is also compiled to assembly and
run through tools/asm_hazard_check.py (VALU-writes-SGPR -> VALU-reads hazards behind
inline asm, which nothing pads automatically): a build with findings is refused,
so an unsafe variant of the field arithmetic cannot ship."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libcuhe_hip.so")
# (object name, source, extra flags)
UNITS = [("cuhe_context", "cuhe_context.hip", []),
         ("cuhe_transforms", "cuhe_transforms.hip", []),
         ("cuhe_keyswitch", "cuhe_keyswitch.hip", []),
         ("icrt_mfma", "icrt_mfma.hip", []),
         ("ntt_onewg_12", "ntt_onewg_inst.hip", ["-DCUHE_OW_LGH=12"]),
         ("ntt_onewg_13", "ntt_onewg_inst.hip", ["-DCUHE_OW_LGH=13"]),
         ("ntt_onewg_14", "ntt_onewg_inst.hip", ["-DCUHE_OW_LGH=14"]),
         ("ntt_onewg_15", "ntt_onewg_inst.hip", ["-DCUHE_OW_LGH=15"])]
COMMON = ["ntt_kernels.cuh", "modp.cuh", os.path.join("..", "..", "include", "cuhe_hip.h")]
ABI = ["cuhe_internal.hpp", "ops_kernels.cuh", "icrt_mfma.cuh", "host_math.hpp", "comm.hpp", "ntt_onewg.hpp"] + COMMON
DEPS = {"cuhe_context.hip": ["cuhe_context.hip"] + ABI, "cuhe_transforms.hip": ["cuhe_transforms.hip"] + ABI,
        "cuhe_keyswitch.hip": ["cuhe_keyswitch.hip"] + ABI, "icrt_mfma.hip": ["icrt_mfma.hip"] + ABI,
        "ntt_onewg_inst.hip": ["ntt_onewg_inst.hip", "ntt_onewg.cuh", "ntt_onewg.hpp"] + COMMON}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden", "-Wno-unused-value", "-Wno-unused-command-line-argument"]


def _newer(target, src):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(os.path.join(CSRC, d)) and os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS[src])


def stale():
    return not os.path.exists(LIB) or any(_newer(LIB, src) for _, src, _ in UNITS)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    if not force and not stale():
        return LIB
    todo = [(name, src, extra) for name, src, extra in UNITS
            if force or _newer(os.path.join(OBJDIR, name + ".o"), src) or _newer(os.path.join(OBJDIR, name + ".s"), src)]
    procs = []
    for name, src, extra in todo:
        path = os.path.join(CSRC, src)
        for kind, args in (("s", ["--cuda-device-only", "-S"]), ("o", ["-fPIC", "-c"])):
            cmd = [HIPCC] + FLAGS + extra + args + ["-o", os.path.join(OBJDIR, name + "." + kind + ".tmp"), path]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((name, kind, subprocess.Popen(cmd)))
    failed = [(n, k) for n, k, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("compilation failed: %s" % failed)
    for name, kind, _ in procs:
        if kind != "s":
            continue
        asm = os.path.join(OBJDIR, name + ".s.tmp")
        chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_check.py"), "--asm", asm], capture_output=True, text=True)
        if verbose or chk.returncode:
            print(name + ": " + chk.stdout.strip()[-3000:], flush=True)
        if chk.returncode:
            raise RuntimeError("inline-asm hazard check failed in %s: the library is not built" % name)
    for name, kind, _ in procs:
        os.replace(os.path.join(OBJDIR, name + "." + kind + ".tmp"), os.path.join(OBJDIR, name + "." + kind))
    cmd = [HIPCC] + FLAGS + ["-fPIC", "-shared"] + [os.path.join(OBJDIR, n + ".o") for n, _, _ in UNITS] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def asm_files():
    """assembly of every translation unit of the last build (tests/test_asm_hazards.py, tools/isa_hist.py)"""
    return [os.path.join(OBJDIR, n + ".s") for n, _, _ in UNITS]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
