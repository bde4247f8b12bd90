"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference hot path (oracle/oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (cuhe_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

P = 0xFFFFFFFF00000001
G = 15893793146607301539


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mSize", "modLen", "modLen2", "rawLen", "crtLen", "nttLen",
        "logCoeffMax", "logCoeffMin", "logCoeffCut",
        "depth", "modMsg", "logMsg", "wordsMsg",
        "logRelin", "numEvalKey", "logCrtPrime", "numCrtPrime")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("oracle.c", "oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u64, u32, i32 = C.c_uint64, C.c_uint32, C.c_int
        vp = C.c_void_p
        L.orc_add_modP.restype = u64; L.orc_add_modP.argtypes = [u64, u64]
        L.orc_sub_modP.restype = u64; L.orc_sub_modP.argtypes = [u64, u64]
        L.orc_mul_modP.restype = u64; L.orc_mul_modP.argtypes = [u64, u64]
        L.orc_ls_modP.restype = u64; L.orc_ls_modP.argtypes = [u64, i32]
        L.orc_pow_modP.restype = u64; L.orc_pow_modP.argtypes = [u64, u64]
        L.orc_len_inv.restype = u64; L.orc_len_inv.argtypes = [i32]
        for f in ("orc_ntt_naive", "orc_ntt_ext", "orc_ntt_full"):
            getattr(L, f).restype = None; getattr(L, f).argtypes = [vp, vp, i32]
        L.orc_ntt_ext_batch.restype = i32; L.orc_ntt_ext_batch.argtypes = [vp, vp, i32, i32, i32]
        L.orc_ntt_ext_fast_batch.restype = i32; L.orc_ntt_ext_fast_batch.argtypes = [vp, vp, i32, i32, i32]
        for f in ("orc_add_modP_div", "orc_mul_modP_div"):
            getattr(L, f).restype = u64; getattr(L, f).argtypes = [u64, u64]
        L.orc_intt_modp.restype = None; L.orc_intt_modp.argtypes = [vp, vp, i32, u32]
        L.orc_set_param.restype = i32
        L.orc_set_param.argtypes = [C.POINTER(Params)] + [i32] * 6
        for f in ("orc_num_crt_prime", "orc_log_coeff", "orc_words_coeff", "orc_num_eval_key", "orc_get_level"):
            getattr(L, f).restype = i32; getattr(L, f).argtypes = [C.POINTER(Params), i32]
        L.orc_is_prime_u32.restype = i32; L.orc_is_prime_u32.argtypes = [u32]
        L.orc_gen_crt_primes.restype = i32; L.orc_gen_crt_primes.argtypes = [C.POINTER(Params), vp]
        L.orc_cyclotomic.restype = i32; L.orc_cyclotomic.argtypes = [i32, vp, i32]
        L.orc_ctx_create.restype = vp; L.orc_ctx_create.argtypes = [i32] * 6 + [vp]
        L.orc_ctx_destroy.restype = None; L.orc_ctx_destroy.argtypes = [vp]
        L.orc_ctx_params.restype = C.POINTER(Params); L.orc_ctx_params.argtypes = [vp]
        L.orc_ctx_primes.restype = C.POINTER(u32); L.orc_ctx_primes.argtypes = [vp]
        L.orc_ctx_invp.restype = C.POINTER(u32); L.orc_ctx_invp.argtypes = [vp]
        L.orc_ctx_coeff_modulus.restype = i32; L.orc_ctx_coeff_modulus.argtypes = [vp, i32, vp, i32]
        sig = {
            "orc_crt": [vp, vp, vp, i32], "orc_icrt": [vp, vp, vp, i32],
            "orc_ntt": [vp, vp, vp, i32], "orc_intt_hold": [vp, vp, vp, i32], "orc_intt": [vp, vp, vp, i32],
            "orc_poly_reduce_exact": [vp, vp, vp, i32], "orc_barrett": [vp, vp, vp, i32],
            "orc_intt_mod": [vp, vp, vp, i32],
            "orc_ntt_mul": [vp, vp, vp, vp, i32], "orc_ntt_add": [vp, vp, vp, vp, i32],
            "orc_ntt_mul_nx1": [vp, vp, vp, vp, i32], "orc_ntt_add_nx1": [vp, vp, vp, vp, i32],
            "orc_crt_add": [vp, vp, vp, vp, i32], "orc_crt_add_int": [vp, vp, vp, C.c_uint, i32],
            "orc_crt_add_nx1": [vp, vp, vp, vp, i32],
            "orc_modswitch": [vp, vp, vp, i32], "orc_nttw": [vp, vp, vp, i32],
            "orc_init_relin": [vp, vp, vp], "orc_relin": [vp, vp, vp, i32, vp],
            "orc_mul_raw": [vp, vp, vp, vp, i32], "orc_mul_relin_crt": [vp, vp, vp, vp, i32, vp],
        }
        for f, a in sig.items():
            getattr(L, f).restype = None; getattr(L, f).argtypes = a
        L.orc_negacyclic_mul_modp_naive.restype = None; L.orc_negacyclic_mul_modp_naive.argtypes = [vp, vp, vp, i32, u32]
        L.orc_negacyclic_mul_modp.restype = i32; L.orc_negacyclic_mul_modp.argtypes = [vp, vp, vp, i32, u32]
        L.orc_nc_relin_modp.restype = i32; L.orc_nc_relin_modp.argtypes = [vp, vp, vp, i32, i32, u32]
        L.orc_nc_mul_relin_crt_batch.restype = i32; L.orc_nc_mul_relin_crt_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp]
        L.orc_gmp_available.restype = i32; L.orc_gmp_available.argtypes = []
        L.orc_gmp_mul_xn1.restype = i32; L.orc_gmp_mul_xn1.argtypes = [vp, vp, vp, i32, i32, vp, i32]
        L.orc_set_threads.restype = i32; L.orc_set_threads.argtypes = [i32]
        L.orc_fill_u32_below.restype = None; L.orc_fill_u32_below.argtypes = [vp, C.c_size_t, u32, u64]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def splitmix_u32_below(n, bound, seed):
    out = np.empty(n, dtype=np.uint32)
    lib().orc_fill_u32_below(_p(out), n, bound, seed)
    return out


def ntt_naive(x, length):
    x = np.ascontiguousarray(x, dtype=np.uint32)
    out = np.empty(length, dtype=np.uint64)
    lib().orc_ntt_naive(_p(out), _p(x), length)
    return out


def ntt_ext(x, length):
    x = np.ascontiguousarray(x, dtype=np.uint32)
    assert x.size >= length // 2
    out = np.empty(length, dtype=np.uint64)
    lib().orc_ntt_ext(_p(out), _p(x), length)
    return out


def ntt_ext_batch(x, length, threads=0):
    """x: u32[batch][length/2] -> (u64[batch][length], threads used)"""
    x = np.ascontiguousarray(x, dtype=np.uint32)
    batch = x.shape[0]
    out = np.empty((batch, length), dtype=np.uint64)
    used = lib().orc_ntt_ext_batch(_p(out), _p(x), length, batch, threads if threads > 0 else host_cores())
    return out, used


def ntt_ext_fast_batch(x, length, threads=0):
    """the same transforms through the table-sharing throughput form (bench.py cpu_baseline): (u64[batch][length], threads used)"""
    x = np.ascontiguousarray(x, dtype=np.uint32)
    batch = x.shape[0]
    out = np.empty((batch, length), dtype=np.uint64)
    used = lib().orc_ntt_ext_fast_batch(_p(out), _p(x), length, batch, threads if threads > 0 else host_cores())
    if used < 0:
        raise RuntimeError("orc_ntt_ext_fast_batch: no table slot for length %d" % length)
    return out, used


def intt_modp(X, length, p):
    X = np.ascontiguousarray(X, dtype=np.uint64)
    out = np.empty(length, dtype=np.uint32)
    lib().orc_intt_modp(_p(out), _p(X), length, p)
    return out


def negacyclic_mul_modp(a, b, p, naive=False):
    """(a * b mod x^n + 1) mod p for residue rows a, b (u32[n], entries below p): by definition (naive) or through the
    twisted length-n transform (needs 2 n p^2 < P)"""
    a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
    out = np.empty(a.size, dtype=np.uint32)
    if naive:
        lib().orc_negacyclic_mul_modp_naive(_p(out), _p(a), _p(b), a.size, p)
    else:
        assert lib().orc_negacyclic_mul_modp(_p(out), _p(a), _p(b), a.size, p) == 0, "2 n p^2 >= P"
    return out


def nc_relin_modp(win, key, p):
    """sum_j win[j] * key[j] mod (x^n + 1) mod p; win, key: u32[k][n]"""
    win = np.ascontiguousarray(win, dtype=np.uint32); key = np.ascontiguousarray(key, dtype=np.uint32)
    k, n = win.shape
    out = np.empty(n, dtype=np.uint32)
    assert lib().orc_nc_relin_modp(_p(out), _p(win), _p(key), k, n, p) == 0
    return out


def gmp_mul_xn1(a_raw, b_raw, q):
    """(a * b mod x^n + 1) mod q through GMP (Kronecker substitution, one mpz_mul), or None when libgmp is absent;
    a_raw, b_raw: u32[n][W]"""
    a_raw = np.ascontiguousarray(a_raw, dtype=np.uint32); b_raw = np.ascontiguousarray(b_raw, dtype=np.uint32)
    n, W = a_raw.shape
    qW = (q.bit_length() + 31) // 32
    qw = np.frombuffer(q.to_bytes(4 * qW, "little"), dtype=np.uint32).copy()
    out = np.zeros((n, W), dtype=np.uint32)
    if lib().orc_gmp_mul_xn1(_p(out), _p(a_raw), _p(b_raw), n, W, _p(qw), qW) != 0:
        return None
    return out


def host_cores():
    """the CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 logical CPUs
    under a quota of 16: 128 OpenMP threads there run 2x SLOWER than 16, profiles/r06_cpu_thread_scaling.txt)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return max(1, n)


def set_threads(n):
    """OpenMP threads of the Ctx stage loops (0 = every CPU this process may use, host_cores()); returns the count in effect"""
    return lib().orc_set_threads(n if n > 0 else host_cores())


def set_param(d, p, w, mn, cut, m):
    q = Params()
    lib().orc_set_param(C.byref(q), d, p, w, mn, cut, m)
    return q


def gen_crt_primes(q):
    out = np.zeros(q.numCrtPrime, dtype=np.uint32)
    lib().orc_gen_crt_primes(C.byref(q), _p(out))
    return out


def cyclotomic(m):
    buf = np.zeros(m + 2, dtype=np.int32)
    deg = lib().orc_cyclotomic(m, _p(buf), buf.size)
    assert deg >= 0
    return buf[:deg + 1].copy()


class Ctx:
    """orc_ctx wrapper; numpy in / numpy out."""

    def __init__(self, d, p, w, mn, cut, m, modulus=None):
        L = lib()
        mod = None
        if modulus is not None:
            mod = np.ascontiguousarray(modulus, dtype=np.int32)
        self.h = L.orc_ctx_create(d, p, w, mn, cut, m, _p(mod) if mod is not None else None)
        assert self.h, "oracle ctx creation failed"
        self.prm = L.orc_ctx_params(self.h).contents
        self.primes = np.array([L.orc_ctx_primes(self.h)[i] for i in range(self.prm.numCrtPrime)], dtype=np.uint32)

    def close(self):
        if self.h:
            lib().orc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # level helpers
    def np_(self, lvl):
        return lib().orc_num_crt_prime(C.byref(self.prm), lvl)

    def words(self, lvl):
        return lib().orc_words_coeff(C.byref(self.prm), lvl)

    def logq(self, lvl):
        return lib().orc_log_coeff(C.byref(self.prm), lvl)

    def nkeys(self, lvl):
        return lib().orc_num_eval_key(C.byref(self.prm), lvl)

    def coeff_modulus(self, lvl):
        buf = np.zeros(128, dtype=np.uint32)
        n = lib().orc_ctx_coeff_modulus(self.h, lvl, _p(buf), buf.size)
        return int.from_bytes(buf[:n].tobytes(), "little")

    def invp(self):
        n = self.prm.numCrtPrime
        cnt = n * (n - 1) // 2
        ptr = lib().orc_ctx_invp(self.h)
        return np.array([ptr[i] for i in range(cnt)], dtype=np.uint32)

    # stages
    def crt(self, raw, lvl):
        raw = np.ascontiguousarray(raw, dtype=np.uint32)
        out = np.empty((self.np_(lvl), self.prm.crtLen), dtype=np.uint32)
        lib().orc_crt(self.h, _p(out), _p(raw), lvl)
        return out

    def icrt(self, crt, lvl):
        crt = np.ascontiguousarray(crt, dtype=np.uint32)
        out = np.empty((self.prm.rawLen, self.words(lvl)), dtype=np.uint32)
        lib().orc_icrt(self.h, _p(out), _p(crt), lvl)
        return out

    def ntt(self, crt):
        crt = np.ascontiguousarray(crt, dtype=np.uint32)
        n = crt.shape[0]
        out = np.empty((n, self.prm.nttLen), dtype=np.uint64)
        lib().orc_ntt(self.h, _p(out), _p(crt), n)
        return out

    def _n2c(self, fn, X, width):
        X = np.ascontiguousarray(X, dtype=np.uint64)
        n = X.shape[0]
        out = np.empty((n, width), dtype=np.uint32)
        getattr(lib(), fn)(self.h, _p(out), _p(X), n)
        return out

    def intt_hold(self, X):
        return self._n2c("orc_intt_hold", X, self.prm.nttLen)

    def intt(self, X):
        return self._n2c("orc_intt", X, self.prm.crtLen)

    def intt_mod(self, X):
        return self._n2c("orc_intt_mod", X, self.prm.crtLen)

    def _reduce(self, fn, hold):
        hold = np.ascontiguousarray(hold, dtype=np.uint32)
        n = hold.shape[0]
        out = np.empty((n, self.prm.crtLen), dtype=np.uint32)
        getattr(lib(), fn)(self.h, _p(out), _p(hold), n)
        return out

    def poly_reduce_exact(self, hold):
        return self._reduce("orc_poly_reduce_exact", hold)

    def barrett(self, hold):
        return self._reduce("orc_barrett", hold)

    def _bin64(self, fn, x, y):
        x = np.ascontiguousarray(x, dtype=np.uint64); y = np.ascontiguousarray(y, dtype=np.uint64)
        out = np.empty_like(x)
        getattr(lib(), fn)(self.h, _p(out), _p(x), _p(y), x.shape[0])
        return out

    def ntt_mul(self, x, y): return self._bin64("orc_ntt_mul", x, y)
    def ntt_add(self, x, y): return self._bin64("orc_ntt_add", x, y)
    def ntt_mul_nx1(self, x, s): return self._bin64("orc_ntt_mul_nx1", x, s)
    def ntt_add_nx1(self, x, s): return self._bin64("orc_ntt_add_nx1", x, s)

    def crt_add(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.uint32); y = np.ascontiguousarray(y, dtype=np.uint32)
        out = x.copy()
        lib().orc_crt_add(self.h, _p(out), _p(x), _p(y), x.shape[0])
        return out

    def crt_add_int(self, x, a):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        out = x.copy()
        lib().orc_crt_add_int(self.h, _p(out), _p(x), a, x.shape[0])
        return out

    def crt_add_nx1(self, x, s):
        x = np.ascontiguousarray(x, dtype=np.uint32); s = np.ascontiguousarray(s, dtype=np.uint32)
        out = x.copy()
        lib().orc_crt_add_nx1(self.h, _p(out), _p(x), _p(s), x.shape[0])
        return out

    def modswitch(self, x):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        n = x.shape[0]
        out = np.zeros((n - 1, self.prm.crtLen), dtype=np.uint32)
        lib().orc_modswitch(self.h, _p(out), _p(x), n)
        return out

    def nttw(self, raw, lvl):
        raw = np.ascontiguousarray(raw, dtype=np.uint32)
        out = np.empty((self.nkeys(lvl), self.prm.nttLen), dtype=np.uint64)
        lib().orc_nttw(self.h, _p(out), _p(raw), lvl)
        return out

    def init_relin(self, evalkey_raw):
        ek_raw = np.ascontiguousarray(evalkey_raw, dtype=np.uint32)
        out = np.empty((self.prm.numCrtPrime, self.prm.numEvalKey, self.prm.nttLen), dtype=np.uint64)
        lib().orc_init_relin(self.h, _p(out), _p(ek_raw))
        return out

    def relin(self, raw, lvl, ek):
        raw = np.ascontiguousarray(raw, dtype=np.uint32)
        out = np.empty((self.np_(lvl), self.prm.nttLen), dtype=np.uint64)
        lib().orc_relin(self.h, _p(out), _p(raw), lvl, _p(ek))
        return out

    def mul_raw(self, a, b, lvl):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        out = np.empty((self.prm.rawLen, self.words(lvl)), dtype=np.uint32)
        lib().orc_mul_raw(self.h, _p(out), _p(a), _p(b), lvl)
        return out

    def mul_relin_crt(self, a, b, lvl, ek):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        out = np.empty((self.np_(lvl), self.prm.crtLen), dtype=np.uint32)
        lib().orc_mul_relin_crt(self.h, _p(out), _p(a), _p(b), lvl, _p(ek))
        return out


    def key_residues(self, evalkey_raw):
        """u32[K][np0][crtLen]: the CRT rows of the K raw evaluation keys (Relinearization.cu:43-57 before the transforms)"""
        ek_raw = np.ascontiguousarray(evalkey_raw, dtype=np.uint32)
        return np.stack([self.crt(ek_raw[j], 0) for j in range(ek_raw.shape[0])])

    def nc_prepare(self, lvl, ekc):
        """keys transformed once (bench.py's CPU leg for mul + relin): a handle for nc_mul_relin_prepared / nc_prepared_free"""
        L = lib()
        L.orc_nc_prepare.restype = C.c_void_p; L.orc_nc_prepare.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_nc_prepared_free.restype = None; L.orc_nc_prepared_free.argtypes = [C.c_void_p]
        L.orc_nc_mul_relin_prepared.restype = C.c_int; L.orc_nc_mul_relin_prepared.argtypes = [C.c_void_p] * 4
        ekc = np.ascontiguousarray(ekc, dtype=np.uint32)
        assert ekc.shape[1:] == (self.prm.numCrtPrime, self.prm.crtLen)
        h = L.orc_nc_prepare(self.h, lvl, _p(ekc))
        assert h, "not a ring x^n + 1 within the lift bounds"
        return h

    def nc_mul_relin_prepared(self, h, a, b, lvl):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        assert a.shape == b.shape == (self.np_(lvl), self.prm.crtLen)
        out = np.empty_like(a)
        assert lib().orc_nc_mul_relin_prepared(h, _p(out), _p(a), _p(b)) == 0
        return out

    def nc_prepared_free(self, h):
        lib().orc_nc_prepared_free(h)

    def nc_mul_relin_crt_batch(self, a, b, lvl, ekc):
        """cAnd + relin of B pairs on a ring x^n + 1 (a, b: u32[B][np][crtLen]), per prime through the negacyclic
        restatement: the form of mul_relin_crt that fits BASELINE config 4 (no np x K x nttLen key table)"""
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        ekc = np.ascontiguousarray(ekc, dtype=np.uint32)
        assert a.shape == b.shape == (a.shape[0], self.np_(lvl), self.prm.crtLen) and ekc.shape[1:] == (self.prm.numCrtPrime, self.prm.crtLen)
        out = np.empty_like(a)
        assert lib().orc_nc_mul_relin_crt_batch(self.h, _p(out), _p(a), _p(b), a.shape[0], lvl, _p(ekc)) == 0, "not a ring x^n + 1 within the lift bounds"
        return out


# ---- big-int <-> raw layout helpers (cuhe/CuHE.cu:317-348 z2r / r2z) ----
def ints_to_raw(vals, rawlen, words):
    """list of non-negative python ints -> u32[rawLen][W] little-endian words."""
    out = np.zeros((rawlen, words), dtype=np.uint32)
    for i, v in enumerate(vals):
        assert 0 <= v < (1 << (32 * words))
        out[i] = np.frombuffer(int(v).to_bytes(4 * words, "little"), dtype=np.uint32)
    return out


def raw_to_ints(raw, count=None):
    raw = np.ascontiguousarray(raw, dtype=np.uint32)
    n = raw.shape[0] if count is None else count
    return [int.from_bytes(raw[i].tobytes(), "little") for i in range(n)]


def random_raw(rawlen, modlen, words, q, seed):
    """uniform-ish big coefficients in [0, q) for idx < modLen, zero above."""
    nw = words + 1
    r = splitmix_u32_below(modlen * nw, 0xFFFFFFFF, seed).reshape(modlen, nw)
    vals = [int.from_bytes(r[i].tobytes(), "little") % q for i in range(modlen)]
    return ints_to_raw(vals, rawlen, words), vals
