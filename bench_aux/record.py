"""Constants of the measurement record and the hash that ties a committed counter record to the kernel sources it was measured on."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
RING_PARAMS = {"2^15": (25, 2, 16, 576, 24, 65536),       # x^32768 + 1: 48 primes < 2^24, 72 keys (the reference's largest ring)
               "2^16": (25, 2, 16, 552, 23, 131072)}      # x^65536 + 1: BASELINE config 4 read literally, 48 primes < 2^23, 69 keys


def code_only(text):
    """a source text without comments and blank space: what the hash below is taken over, so that editing a comment does not
    orphan a measurement"""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    lines = []
    for line in text.split("\n"):
        line = re.sub(r"\s+", " ", re.sub(r"//.*$", "", line)).strip()
        if line:
            lines.append(line)
    return "\n".join(lines)


def kernel_sha16():
    """identifies the transform kernels a committed PMC figure was measured on (code of the three kernel headers, comments removed)"""
    h = hashlib.sha256()
    for f in ("modp.cuh", "ntt_kernels.cuh", "ntt_onewg.cuh"):
        h.update(code_only(open(os.path.join(ROOT, "cuhe_amd", "csrc", f)).read()).encode())
    return h.hexdigest()[:16]
