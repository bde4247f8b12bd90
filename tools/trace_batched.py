"""Only the batched multiply + relinearise call of bench.py (BASELINE config 4 shape), for a kernel trace:
    rocprofv3 --kernel-trace --stats -d out -o s -- python tools/trace_batched.py [batch] [calls] [ring]
(the evaluation-key set-up adds 72 small transform calls to the same kernel names; `calls` batched calls dominate)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from cuhe_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ring = sys.argv[3] if len(sys.argv) > 3 else "2^15"
lib = capi.lib


def ck(rc):
    if rc != 0:
        raise RuntimeError(lib.cuhe_hip_last_error().decode())


dev = torch.device("cuda:0")
d, p, w, mn, cut, m = bench.RING_PARAMS[ring]
ck(lib.cuhe_hip_set_negacyclic(-1))
ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
ck(lib.cuhe_hip_init(None, 0))
q = capi.get_params()
npn, L, K, W = q.numCrtPrime, lib.cuhe_hip_ct_len(), q.numEvalKey, lib.cuhe_hip_words_coeff(0)
rng = np.random.default_rng(7)
ek = rng.integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32)
ek[:, :, W - 1] &= 0x7FFF
ck(lib.cuhe_hip_init_relin(ek.ctypes.data_as(C.c_void_p)))
gen = torch.Generator(device=dev); gen.manual_seed(5)
na = torch.randint(0, 1 << 62, (B * npn, L), dtype=torch.int64, device=dev, generator=gen)
nb = torch.randint(0, 1 << 62, (B * npn, L), dtype=torch.int64, device=dev, generator=gen)
out = torch.empty((B * npn, q.crtLen), dtype=torch.int32, device=dev)
import time


def mark():            # CUHE_TRACE_MARK=1: the library's probe kernel around the timed calls (tools/rocpd_summary.py --between k_probe_valu)
    if os.environ.get("CUHE_TRACE_MARK"):
        x = [C.c_double() for _ in range(3)]
        torch.cuda.synchronize()
        ck(lib.cuhe_hip_probe_valu(0, 1, 1, C.byref(x[0]), C.byref(x[1]), C.byref(x[2])))


for i in range(calls + 2):
    if i == 2:
        mark()
        torch.cuda.synchronize(); t0 = time.perf_counter()
    ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), 0, B, 0, None))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
mark()
print("batch %d ring %s: %.4f ms per ciphertext" % (B, ring, dt / calls / B * 1e3))
