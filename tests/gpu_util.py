"""Helpers for the -m gpu parity tests: device buffers via torch (plumbing only),
every compute call goes through the C ABI (cuhe_amd.capi)."""
import ctypes as C

import numpy as np
import torch

from cuhe_amd import capi

lib = capi.lib
ck = capi.check
DEV = torch.device("cuda:0")


def to_dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).to(DEV)
    if a.dtype == np.uint64:
        return torch.from_numpy(a.view(np.int64)).to(DEV)
    raise TypeError(a.dtype)


def empty_u32(*shape):
    return torch.zeros(shape, dtype=torch.int32, device=DEV)


def empty_u64(*shape):
    return torch.zeros(shape, dtype=torch.int64, device=DEV)


def host_u32(t):
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint32)


def host_u64(t):
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint64)


class GpuCtx:
    """(re)initialises the library-global context for one parameter set."""

    def __init__(self, d, p, w, mn, cut, m, modulus=None, negacyclic=True):
        """negacyclic=False: keep the reference's cyclic representation even where the negacyclic one applies."""
        lib.cuhe_hip_shutdown()
        lib.cuhe_hip_reset_parameters()
        ck(lib.cuhe_hip_set_negacyclic(-1 if negacyclic else 0))
        ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
        if modulus is None:
            ck(lib.cuhe_hip_init(None, 0))
        else:
            mod = np.ascontiguousarray(modulus, dtype=np.int32)
            ck(lib.cuhe_hip_init(mod.ctypes.data_as(C.c_void_p), mod.size))
        self.prm = capi.get_params()
        pr = np.zeros(self.prm.numCrtPrime, dtype=np.uint32)
        ck(lib.cuhe_hip_get_crt_primes(pr.ctypes.data_as(C.c_void_p), pr.size))
        self.primes = pr
        self.nc = bool(lib.cuhe_hip_ct_negacyclic())        # ciphertext-domain rows are negacyclic transforms of modLen points
        self.ctlen = lib.cuhe_hip_ct_len()

    def close(self):
        lib.cuhe_hip_shutdown()
        lib.cuhe_hip_reset_parameters()
        lib.cuhe_hip_set_negacyclic(-1)

    # --- ciphertext-domain transforms (what CuCtxt uses): rows of ctlen
    def ct_ntt(self, crt, lvl):
        d_in = to_dev(crt)
        d_out = empty_u64(self.np_(lvl), self.ctlen)
        ck(lib.cuhe_hip_ct_ntt(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u64(d_out)

    def ct_intt(self, X, lvl, prod):
        d_in = to_dev(X)
        d_out = empty_u32(self.np_(lvl), self.prm.crtLen)
        ck(lib.cuhe_hip_ct_intt(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 1 if prod else 0, 0, None))
        return host_u32(d_out)

    def ct_mul(self, x, y, lvl): return self._bin64("cuhe_hip_ct_mul", x, y, lvl)
    def ct_add(self, x, y, lvl): return self._bin64("cuhe_hip_ct_add", x, y, lvl)

    def relin_crt(self, raw, lvl):
        """CuCtxt::relin on a raw ciphertext: key switch, then n2c of the sums -> u32[np][crtLen]"""
        return self.ct_intt(self.relin(raw, lvl), lvl, True)

    def np_(self, lvl): return lib.cuhe_hip_num_crt_prime(lvl)
    def words(self, lvl): return lib.cuhe_hip_words_coeff(lvl)
    def logq(self, lvl): return lib.cuhe_hip_log_coeff(lvl)
    def nkeys(self, lvl): return lib.cuhe_hip_num_eval_key(lvl)

    def coeff_modulus(self, lvl):
        buf = (C.c_uint8 * 4096)()
        n = C.c_size_t(0)
        ck(lib.cuhe_hip_get_coeff_modulus(lvl, buf, 4096, C.byref(n)))
        return int.from_bytes(bytes(buf[:n.value]), "little")

    def crt_primes(self):
        out = (C.c_uint32 * self.prm.numCrtPrime)()
        ck(lib.cuhe_hip_get_crt_primes(out, self.prm.numCrtPrime))
        return [int(x) for x in out]

    # --- stages: numpy in, numpy out, compute on the GPU through the C ABI
    def crt(self, raw, lvl):
        d_in = to_dev(raw)
        d_out = empty_u32(self.np_(lvl), self.prm.crtLen)
        ck(lib.cuhe_hip_crt(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(d_out)

    def icrt(self, crt, lvl):
        d_in = to_dev(crt)
        d_out = empty_u32(self.prm.rawLen, self.words(lvl))
        ck(lib.cuhe_hip_icrt(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(d_out)

    def ntt(self, crt, lvl):
        d_in = to_dev(crt)
        d_out = empty_u64(self.np_(lvl), self.prm.nttLen)
        ck(lib.cuhe_hip_ntt(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u64(d_out)

    def intt(self, X, lvl):
        d_in = to_dev(X)
        d_out = empty_u32(self.np_(lvl), self.prm.crtLen)
        ck(lib.cuhe_hip_intt(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(d_out)

    def intt_double_deg(self, X, lvl):
        d_in = to_dev(X)
        d_out = empty_u32(self.np_(lvl), self.prm.nttLen)
        ck(lib.cuhe_hip_intt_double_deg(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(d_out)

    def intt_mod(self, X, lvl):
        d_in = to_dev(X)
        d_out = empty_u32(self.np_(lvl), self.prm.crtLen)
        ck(lib.cuhe_hip_intt_mod(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(d_out)

    def barrett(self, hold, lvl):
        d_in = to_dev(hold)
        d_out = empty_u32(self.np_(lvl), self.prm.crtLen)
        ck(lib.cuhe_hip_barrett(d_out.data_ptr(), d_in.data_ptr(), lvl, 0, None))
        return host_u32(d_out)

    def _bin64(self, fn, x, y, lvl):
        dx, dy = to_dev(x), to_dev(y)
        dz = empty_u64(*x.shape)
        ck(getattr(lib, fn)(dz.data_ptr(), dx.data_ptr(), dy.data_ptr(), self.logq(lvl), 0, None))
        return host_u64(dz)

    def ntt_mul(self, x, y, lvl): return self._bin64("cuhe_hip_ntt_mul", x, y, lvl)
    def ntt_add(self, x, y, lvl): return self._bin64("cuhe_hip_ntt_add", x, y, lvl)
    def ntt_mul_nx1(self, x, s, lvl): return self._bin64("cuhe_hip_ntt_mul_nx1", x, s, lvl)
    def ntt_add_nx1(self, x, s, lvl): return self._bin64("cuhe_hip_ntt_add_nx1", x, s, lvl)

    def crt_add(self, x, y, lvl):
        dx, dy = to_dev(x), to_dev(y)
        dz = dx.clone()
        ck(lib.cuhe_hip_crt_add(dz.data_ptr(), dx.data_ptr(), dy.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(dz)

    def crt_add_int(self, x, a, lvl):
        dx = to_dev(x)
        dz = dx.clone()
        ck(lib.cuhe_hip_crt_add_int(dz.data_ptr(), dx.data_ptr(), a, self.logq(lvl), 0, None))
        return host_u32(dz)

    def crt_add_nx1(self, x, s, lvl):
        dx, ds = to_dev(x), to_dev(s)
        dz = dx.clone()
        ck(lib.cuhe_hip_crt_add_nx1(dz.data_ptr(), dx.data_ptr(), ds.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(dz)

    def modswitch(self, x, lvl, in_place=False):
        dx = to_dev(x)
        if in_place:                                # CuCtxt::modSwitch calls crtModSwitch(cRep_, cRep_, ...) (CuHE.cu:550)
            ck(lib.cuhe_hip_crt_mod_switch(dx.data_ptr(), dx.data_ptr(), self.logq(lvl), 0, None))
            return host_u32(dx)[:-1]
        dz = empty_u32(self.np_(lvl) - 1, self.prm.crtLen)
        ck(lib.cuhe_hip_crt_mod_switch(dz.data_ptr(), dx.data_ptr(), self.logq(lvl), 0, None))
        return host_u32(dz)

    def nttw(self, raw, lvl):
        d_in = to_dev(raw)
        d_out = empty_u64(self.nkeys(lvl), self.prm.nttLen)
        ck(lib.cuhe_hip_nttw(d_out.data_ptr(), d_in.data_ptr(), self.logq(lvl), 0, None))
        return host_u64(d_out)

    def init_relin(self, ek_raw):
        ek_raw = np.ascontiguousarray(ek_raw, dtype=np.uint32)
        ck(lib.cuhe_hip_init_relin(ek_raw.ctypes.data_as(C.c_void_p)))

    def relin(self, raw, lvl):
        """ct-domain rows u64[np][ctlen] (== the reference's NTT-domain rows on the cyclic representation)"""
        d_in = to_dev(raw)
        d_out = empty_u64(self.np_(lvl), self.ctlen)
        ck(lib.cuhe_hip_relinearization(d_out.data_ptr(), d_in.data_ptr(), lvl, 0, None))
        return host_u64(d_out)

    def mul_raw(self, a_raw, b_raw, lvl, cyclic_api=False):
        """mulZZX at the raw level (cuhe/CuHE.cu:259-268) with device-resident intermediates: through the ciphertext-
        domain entry points CuCtxt uses, or (cyclic_api) through the reference-contract ntt / nttMul / inttMod."""
        logq = self.logq(lvl)
        np_, q = self.np_(lvl), self.prm
        da, db = to_dev(a_raw), to_dev(b_raw)
        ca, cb = empty_u32(np_, q.crtLen), empty_u32(np_, q.crtLen)
        rl = q.nttLen if cyclic_api else self.ctlen
        na, nb = empty_u64(np_, rl), empty_u64(np_, rl)
        out = empty_u32(q.rawLen, self.words(lvl))
        ck(lib.cuhe_hip_crt(ca.data_ptr(), da.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_crt(cb.data_ptr(), db.data_ptr(), logq, 0, None))
        if cyclic_api:
            ck(lib.cuhe_hip_ntt(na.data_ptr(), ca.data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ntt(nb.data_ptr(), cb.data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ntt_mul(na.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_intt_mod(ca.data_ptr(), na.data_ptr(), logq, 0, None))
        else:
            ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), ca.data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), cb.data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ct_mul(na.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ct_intt(ca.data_ptr(), na.data_ptr(), logq, 1, 0, None))
        ck(lib.cuhe_hip_icrt(out.data_ptr(), ca.data_ptr(), logq, 0, None))
        return host_u32(out)

    def mul_relin_crt(self, a_crt, b_crt, lvl, fused=False):
        """cAnd + relin (cuhe/CuHE.cu:101,570-581), operands and result in the CRT domain.  fused: relinearization ; n2c as the ONE
        call CuCtxt::relin makes since round 5 (cuhe_hip_relin_crt)."""
        logq = self.logq(lvl)
        np_, q = self.np_(lvl), self.prm
        ca, cb = to_dev(a_crt), to_dev(b_crt)
        na, nb = empty_u64(np_, self.ctlen), empty_u64(np_, self.ctlen)
        cr = empty_u32(np_, q.crtLen)
        raw = empty_u32(q.rawLen, self.words(lvl))
        ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), ca.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), cb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_mul(na.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_intt(cr.data_ptr(), na.data_ptr(), logq, 1, 0, None))
        ck(lib.cuhe_hip_icrt(raw.data_ptr(), cr.data_ptr(), logq, 0, None))
        if fused:
            ck(lib.cuhe_hip_relin_crt(cr.data_ptr(), raw.data_ptr(), lvl, 0, None))
            return host_u32(cr)
        ck(lib.cuhe_hip_relinearization(na.data_ptr(), raw.data_ptr(), lvl, 0, None))
        ck(lib.cuhe_hip_ct_intt(cr.data_ptr(), na.data_ptr(), logq, 1, 0, None))
        return host_u32(cr)
