"""One-ciphertext multiply + relinearise chain (cAnd ; relin = x2r, relinearization, n2c) at BASELINE config 4, through raw ctypes so
that OLDER builds of the library can be loaded too (tools/build_variant.py, or a worktree of an earlier round): bisecting the 0.269 ->
0.281 ms drift between BENCH_r02 and BENCH_r03.   usage: single_chain.py <libcuhe_hip.so> [ring: 32768 | 65536] [reps]"""
import ctypes as C, sys, time
import numpy as np
import torch
lib = C.CDLL(sys.argv[1])
ring = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
params = (25, 2, 16, 576, 24, 65536) if ring == 32768 else (25, 2, 16, 552, 23, 131072)
lib.cuhe_hip_last_error.restype = C.c_char_p
def ck(s):
    if s != 0: raise RuntimeError(lib.cuhe_hip_last_error().decode())
class P(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mSize", "modLen", "modLen2", "rawLen", "crtLen", "nttLen", "logCoeffMax", "logCoeffMin", "logCoeffCut", "depth", "modMsg",
                                       "logMsg", "wordsMsg", "logRelin", "numEvalKey", "logCrtPrime", "numCrtPrime")]
vp = C.c_void_p
for f, a in (("cuhe_hip_ct_ntt", [vp, vp, C.c_int, C.c_int, vp]), ("cuhe_hip_ct_mul", [vp, vp, vp, C.c_int, C.c_int, vp]), ("cuhe_hip_ct_intt", [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
             ("cuhe_hip_icrt", [vp, vp, C.c_int, C.c_int, vp]), ("cuhe_hip_relinearization", [vp, vp, C.c_int, C.c_int, vp]), ("cuhe_hip_init_relin", [vp]), ("cuhe_hip_init", [vp, C.c_int])):
    getattr(lib, f).argtypes = a
ck(lib.cuhe_hip_set_parameters(*params)); ck(lib.cuhe_hip_init(None, 0))
q = P(); ck(lib.cuhe_hip_get_parameters(C.byref(q)))
npn, L, K, W = q.numCrtPrime, lib.cuhe_hip_ct_len(), q.numEvalKey, lib.cuhe_hip_words_coeff(0)
rng = np.random.default_rng(7)
ek = rng.integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32); ek[:, :, W - 1] &= 0x7FFF
ck(lib.cuhe_hip_init_relin(ek.ctypes.data_as(vp)))
logq = lib.cuhe_hip_log_coeff(0)
dev = torch.device("cuda:0")
a = torch.randint(0, 1 << (q.logCrtPrime - 1), (npn, q.crtLen), dtype=torch.int32, device=dev)
b = torch.randint(0, 1 << (q.logCrtPrime - 1), (npn, q.crtLen), dtype=torch.int32, device=dev)
na = torch.empty((npn, L), dtype=torch.int64, device=dev); nb = torch.empty_like(na); nc = torch.empty_like(na)
cr = torch.empty((npn, q.crtLen), dtype=torch.int32, device=dev); raw = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)
ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), a.data_ptr(), logq, 0, None)); ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), b.data_ptr(), logq, 0, None))
steps = [("ct_mul", lambda: lib.cuhe_hip_ct_mul(nc.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None)),
         ("ct_intt(prod)", lambda: lib.cuhe_hip_ct_intt(cr.data_ptr(), nc.data_ptr(), logq, 1, 0, None)),
         ("icrt", lambda: lib.cuhe_hip_icrt(raw.data_ptr(), cr.data_ptr(), logq, 0, None)),
         ("relinearization", lambda: lib.cuhe_hip_relinearization(nc.data_ptr(), raw.data_ptr(), 0, 0, None)),
         ("ct_intt(relin)", lambda: lib.cuhe_hip_ct_intt(cr.data_ptr(), nc.data_ptr(), logq, 1, 0, None))]
def chain():
    for _, f in steps: ck(f())
for _ in range(5): chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): chain()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / reps
parts = []
for name, f in steps:                      # each step alone, back to back (device time + launch, no overlap between steps)
    for _ in range(3): ck(f())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ck(f())
    torch.cuda.synchronize(); parts.append("%s %.1f" % (name, (time.perf_counter() - t0) / reps * 1e6))
print("%-28s ring %d: chain %.1f us | %s" % (sys.argv[1].split("/")[-1], ring, tot * 1e6, " | ".join(parts)), flush=True)
