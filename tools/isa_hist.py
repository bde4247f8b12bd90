#!/usr/bin/env python3
"""Static instruction histogram of the device code: tools/isa_hist.py [kernel-name-substring ...]
(hipcc --cuda-device-only -S of cuhe_amd/csrc/cuhe_transforms.hip; counts per kernel, VALU per point for the 16-points-per-thread NTT kernels)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pats = sys.argv[1:] or ["ntt_pass1wILi16ELi0", "ntt_pass2wILi16ELi0"]
with tempfile.TemporaryDirectory() as d:
    s = os.path.join(d, "dev.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only",
                           "-S", "-o", s, os.path.join(ROOT, "cuhe_amd/csrc/cuhe_transforms.hip"), "-I" + os.path.join(ROOT, "include")])
    cur, H = None, {}
    for line in open(s):
        m = re.match(r"^(_Z\w+):", line)
        if m: cur = m.group(1); H[cur] = collections.Counter(); continue
        if line.startswith(".Lfunc_end"): cur = None
        m = re.match(r"^\s+([a-z][a-z_0-9]+)\s", line)
        if m and cur: H[cur][m.group(1)] += 1
for k, c in H.items():
    if any(p in k for p in pats):
        valu = sum(v for i, v in c.items() if i.startswith("v_"))
        print(k[:60], "total", sum(c.values()), "valu", valu, "per point (16/thread)", valu / 16)
        print("    " + ", ".join(f"{i} {v}" for i, v in c.most_common(28)))
