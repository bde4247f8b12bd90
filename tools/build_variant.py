#!/usr/bin/env python3
"""Builds a VARIANT of libcuhe_hip.so with extra compiler flags into cuhe_amd/lib/<name>.so (objects in a scratch directory), for
A/B runs on one box through CUHE_HIP_LIB (cuhe_amd/capi.py):  tools/build_variant.py libcuhe_hip_chain.so -DCUHE_MULP_CHAIN"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cuhe_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    procs = []
    for unit, src, uextra in B.UNITS:
        o = os.path.join(d, unit + ".o")
        procs.append((o, subprocess.Popen([B.HIPCC] + B.FLAGS + uextra + extra + ["-fPIC", "-c", "-o", o, os.path.join(B.CSRC, src)])))
    for o, p in procs:
        if p.wait() != 0:
            raise SystemExit("compilation failed: " + o)
    out = os.path.join(B.LIBDIR, name)
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-fPIC", "-shared"] + [o for o, _ in procs] + ["-o", out])
    print(out)
