#!/usr/bin/env python3
"""Resolves the object+offset lines of tests/cxx/sample_profiler.hpp's report with addr2line against the objects in cuhe_amd/lib/
(and the system's HIP runtime):  tools/resolve_samples.py < report.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
where = {"libcuHE.so": os.path.join(ROOT, "cuhe_amd/lib/libcuHE.so"), "libcuhe_hip.so": os.path.join(ROOT, "cuhe_amd/lib/libcuhe_hip.so"),
         "test_prince_flow": os.path.join(ROOT, "cuhe_amd/lib/test_prince_flow")}
for line in sys.stdin:
    m = re.search(r"(\S+\.so[.0-9]*|test_\w+)\+0x([0-9a-f]+)\s*$", line)
    if not m:
        print(line.rstrip()); continue
    cnt, obj, off = line[:m.start()].rstrip(), m.group(1), m.group(2)
    path = where.get(obj)
    if not path:
        for d in ("/opt/rocm/lib", "/usr/lib/x86_64-linux-gnu", "/lib/x86_64-linux-gnu"):
            c = os.path.join(d, obj)
            if os.path.exists(c): path = c; break
    name = "?"
    if path and os.path.exists(path):
        r = subprocess.run(["addr2line", "-f", "-C", "-e", path, "0x" + off], capture_output=True, text=True)
        parts = r.stdout.split("\n")
        name = parts[0][:90] + ("  " + os.path.basename(parts[1]) if len(parts) > 1 and parts[1] and not parts[1].startswith("??") else "")
    print("%s %s+0x%s  %s" % (cnt, obj, off, name))
