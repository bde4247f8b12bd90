"""The bundle-size table of doc/Perf_NTT.txt on this GPU, and the dispatch report of the last transform call."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

from .record import HBM_PEAK_GBS, ROOT


def perf_table(lib, ck, torch, dev, path):
    """doc/Perf_NTT.txt on this GPU: ms per single forward transform when `bundle` transforms share a launch pair, bundle
    1 ... 512, lengths 16K / 32K / 64K, 1024 transforms per measurement on consecutive slabs (tests/test_ntt.cu:67-100,140-151)."""
    cnt = 1024
    ref = {1: (0.0486284, 0.051598, 0.064822), 512: (0.00407564, 0.00804859, 0.0226647)}     # doc/Perf_NTT.txt:5,14
    rows = []
    lens = (16384, 32768, 65536)
    for L in lens:
        ck(lib.cuhe_hip_ntt_prepare(L, 0))
    src = {L: torch.randint(-(1 << 31), (1 << 31) - 1, (cnt, L // 2), dtype=torch.int32, device=dev) for L in lens}
    dst = {L: torch.empty((cnt, L), dtype=torch.int64, device=dev) for L in lens}
    for bundle in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        row = [bundle]
        for L in lens:
            s, d = src[L], dst[L]

            def run():
                for b0 in range(0, cnt, bundle):
                    ck(lib.cuhe_hip_ntt_fwd_batched(d[b0:].data_ptr(), s[b0:].data_ptr(), L, bundle, L // 2, 0, None))
            run(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); run(); torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / cnt * 1e3)
            row.append(best)
        rows.append(row)
    with open(path, "w") as f:
        f.write("# forward NTT (u32 half-length input -> u64 output), ms per single transform with `Num` transforms per launch pair;\n")
        f.write("# 1024 transforms per measurement on consecutive slabs, host launch loop + device time, best of 3 (bench.py --perf-table);\n")
        f.write("# the shape of the reference's doc/Perf_NTT.txt (tests/test_ntt.cu:140-151; its hardware is not stated):\n")
        f.write("#   reference bundle 1:   16K %.7f  32K %.7f  64K %.7f\n#   reference bundle 512: 16K %.7f  32K %.7f  64K %.7f\n" % (ref[1] + ref[512]))
        f.write("%-6s %-14s %-14s %-14s\n" % ("Num", "16K", "32K", "64K"))
        for r in rows:
            f.write("%-6d %-14.7f %-14.7f %-14.7f\n" % tuple(r))
    print(open(path).read())


def dispatch_info(lib):
    """which kernel form the last transform call of this thread took, and the rendezvous give-up count (cuhe_hip_last_dispatch_info)"""
    buf = C.create_string_buffer(256)
    return buf.value.decode() if lib.cuhe_hip_last_dispatch_info(0, buf, 256) == 0 else None
