"""Experiment: does running batched multiply+relinearise calls from several host threads (own stream, own scratch each)
overlap the HBM-bound inner product of one call with the instruction-bound transforms of another?
usage: python tools/exp_relin_overlap.py   (config 4 shape, prints ms per ciphertext for T threads x batch B)"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuhe_amd import capi
lib, ck = capi.lib, capi.check
dev = torch.device("cuda", 0)
d, p, w, mn, cut, m = 25, 2, 16, 576, 24, 65536
lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m)); ck(lib.cuhe_hip_init(None, 0))
q = capi.get_params()
npn, L, K, W = q.numCrtPrime, q.nttLen, q.numEvalKey, lib.cuhe_hip_words_coeff(0)
rng = np.random.default_rng(7)
ek = rng.integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32); ek[:, :, W - 1] &= 0x7FFF
ck(lib.cuhe_hip_init_relin(ek.ctypes.data_as(C.c_void_p)))
logq = lib.cuhe_hip_log_coeff(0)
gen = torch.Generator(device=dev); gen.manual_seed(5)
a = torch.randint(0, 1 << 24, (npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
b = torch.randint(0, 1 << 24, (npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
na = torch.empty((npn, L), dtype=torch.int64, device=dev); nb = torch.empty_like(na)
ck(lib.cuhe_hip_ntt(na.data_ptr(), a.data_ptr(), logq, 0, None)); ck(lib.cuhe_hip_ntt(nb.data_ptr(), b.data_ptr(), logq, 0, None))
torch.cuda.synchronize()

def run(T, B, reps):
    bufs = []
    for t in range(T):
        st = C.c_void_p(); ck(lib.cuhe_hip_stream_create(0, C.byref(st)))
        bufs.append((st, na.repeat(B, 1).contiguous(), nb.repeat(B, 1).contiguous(), torch.empty((B * npn, q.crtLen), dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    def work(t, n):
        st, x, y, o = bufs[t]
        for _ in range(n):
            ck(lib.cuhe_hip_mul_relin_batch(o.data_ptr(), x.data_ptr(), y.data_ptr(), 0, B, 0, st))
        ck(lib.cuhe_hip_stream_sync(0, st))
    for t in range(T): work(t, 1)
    th = [threading.Thread(target=work, args=(t, reps)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    ref = bufs[0][3][:npn].clone()
    same = all(torch.equal(bf[3][:npn], ref) and torch.equal(bf[3][(B - 1) * npn:], ref) for bf in bufs)
    for bf in bufs: ck(lib.cuhe_hip_stream_destroy(0, bf[0]))
    print("threads %d x batch %d: %.4f ms per ciphertext  (results identical: %s)" % (T, B, dt / (T * B * reps) * 1e3, same), flush=True)

for lanes, T, B in ((1, 1, 8), (2, 1, 8), (1, 1, 16), (2, 1, 16), (3, 1, 12), (3, 1, 24), (4, 1, 16), (4, 1, 32), (3, 1, 36), (1, 1, 24)):
    ck(lib.cuhe_hip_set_relin_lanes(lanes))
    print("lanes %d: " % lanes, end="")
    run(T, B, 6)
