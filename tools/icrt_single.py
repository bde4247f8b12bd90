"""Time of ONE inverse CRT call (cuhe_hip_icrt: residues -> raw words) and of a batched call, both kernel forms
(cuhe_hip_set_icrt_mfma), BASELINE config 4 rings:  python tools/icrt_single.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from cuhe_amd import capi

lib = capi.lib


def ck(rc):
    if rc != 0:
        raise RuntimeError(lib.cuhe_hip_last_error().decode())


dev = torch.device("cuda:0")
for ring in ("2^15", "2^16"):
    d, p, w, mn, cut, m = bench.RING_PARAMS[ring]
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    ck(lib.cuhe_hip_set_negacyclic(-1))
    ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
    ck(lib.cuhe_hip_init(None, 0))
    q = capi.get_params()
    for lvl in (0, 12, 24):
        npn, W, logq = lib.cuhe_hip_num_crt_prime(lvl), lib.cuhe_hip_words_coeff(lvl), lib.cuhe_hip_log_coeff(lvl)
        src = torch.randint(0, 1 << 22, (npn, q.crtLen), dtype=torch.int32, device=dev)
        dst = torch.empty((q.rawLen, W), dtype=torch.int32, device=dev)
        res = {}
        for on in (1, 0):
            ck(lib.cuhe_hip_set_icrt_mfma(on))
            for _ in range(5):
                ck(lib.cuhe_hip_icrt(dst.data_ptr(), src.data_ptr(), logq, 0, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                ck(lib.cuhe_hip_icrt(dst.data_ptr(), src.data_ptr(), logq, 0, None))
            e1.record(); torch.cuda.synchronize()
            res[on] = e0.elapsed_time(e1) / 200 * 1e3
            res[(on, "out")] = dst.clone()
        assert torch.equal(res[(1, "out")], res[(0, "out")])
        print("ring %s level %2d (%d primes, %d words): one call, matrix cores %.1f us, VALU %.1f us" % (ring, lvl, npn, W, res[1], res[0]))
    ck(lib.cuhe_hip_set_icrt_mfma(1))
