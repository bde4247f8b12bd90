"""CPU: the oracle's restatement of products modulo x^n + 1 (oracle/oracle.c: orc_negacyclic_mul_modp*, orc_nc_*), the
meaning of the reference's reduction chain (cuhe/Operations.cu:460-501) on rings with m = 2n a power of two and what the
MI355X backend computes there with negacyclic transforms.  Pinned three ways: Python integers by definition, the
O(n^2) C restatement, and the oracle's own cyclic path (zero-padded transforms + exact remainder), which the golden
fixtures pin to (a*b mod Phi_m) mod q (tests/golden/pipeline_pow2_16384.json)."""
import numpy as np
import pytest

import oracle_lib as O


def _py_negacyclic(a, b, p):
    n = len(a)
    c = [0] * n
    for i in range(n):
        for j in range(n):
            k = i + j
            if k < n: c[k] += int(a[i]) * int(b[j])
            else: c[k - n] -= int(a[i]) * int(b[j])
    return np.array([x % p for x in c], dtype=np.uint32)


def test_definition_vs_python_integers():
    rng = np.random.default_rng(1)
    for p in (2097143, 16777213, 33554393):
        a = rng.integers(0, p, 48, dtype=np.uint32); b = rng.integers(0, p, 48, dtype=np.uint32)
        a[0], b[0] = p - 1, p - 1
        pad = lambda v: np.concatenate([v, np.zeros(16, dtype=np.uint32)])          # n = 64
        assert np.array_equal(O.negacyclic_mul_modp(pad(a), pad(b), p, naive=True), _py_negacyclic(pad(a), pad(b), p))


@pytest.mark.parametrize("n,p", [(2048, 16777213), (4096, 8388593), (2048, 33554393)])
def test_transform_form_equals_definition(n, p):
    rng = np.random.default_rng(n + p)
    a = rng.integers(0, p, n, dtype=np.uint32); b = rng.integers(0, p, n, dtype=np.uint32)
    assert np.array_equal(O.negacyclic_mul_modp(a, b, p), O.negacyclic_mul_modp(a, b, p, naive=True))
    full = np.full(n, p - 1, dtype=np.uint32)                                      # largest magnitudes: the centred lift
    assert np.array_equal(O.negacyclic_mul_modp(full, full, p), O.negacyclic_mul_modp(full, full, p, naive=True))
    one = np.zeros(n, dtype=np.uint32); one[0] = 1
    assert np.array_equal(O.negacyclic_mul_modp(a, one, p), a)
    xk = np.zeros(n, dtype=np.uint32); xk[n - 1] = 1                                # x^(n-1): rotation with sign change
    want = np.concatenate([(p - a[1:]) % p, a[:1]]).astype(np.uint32)
    assert np.array_equal(O.negacyclic_mul_modp(a, xk, p), want)


def test_bound_is_enforced():
    n, p = 65536, 16777213          # 2 n p^2 >= P
    a = np.zeros(n, dtype=np.uint32)
    out = np.empty(n, dtype=np.uint32)
    assert O.lib().orc_negacyclic_mul_modp(O._p(out), O._p(a), O._p(a), n, p) == -1


def test_equals_cyclic_path_of_the_oracle():
    """x^8192 + 1: the oracle's reference-shaped path (zero-padded 16K-point transforms, product, inverse, exact
    remainder) against the negacyclic restatement, per CRT prime"""
    o = O.Ctx(3, 2, 16, 50, 25, 16384)
    try:
        q, npr = o.prm, o.np_(0)
        rng = np.random.default_rng(5)
        a = np.zeros((npr, q.crtLen), dtype=np.uint32); b = np.zeros_like(a)
        for i in range(npr):
            a[i, :q.modLen] = rng.integers(0, o.primes[i], q.modLen); b[i, :q.modLen] = rng.integers(0, o.primes[i], q.modLen)
        want = o.intt_mod(o.ntt_mul(o.ntt(a), o.ntt(b)))
        for i in range(npr):
            assert np.array_equal(O.negacyclic_mul_modp(a[i], b[i], int(o.primes[i])), want[i]), i
    finally:
        o.close()


def test_key_switch_sum():
    n, k, p, w = 1024, 5, 8388593, 16
    rng = np.random.default_rng(9)
    win = rng.integers(0, 1 << w, (k, n), dtype=np.uint32); key = rng.integers(0, p, (k, n), dtype=np.uint32)
    acc = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        acc = (acc + O.negacyclic_mul_modp((win[j] % p).astype(np.uint32), key[j], p, naive=True)) % p
    assert np.array_equal(O.nc_relin_modp(win, key, p), acc.astype(np.uint32))


def test_degree_65536_parameters_match_the_library():
    """ring degree 2^16 (m = 131072): beyond the reference; library and oracle derive the same parameter set"""
    from cuhe_amd import capi
    q = O.set_param(25, 2, 16, 552, 23, 131072)
    capi.lib.cuhe_hip_reset_parameters()
    capi.check(capi.lib.cuhe_hip_set_parameters(25, 2, 16, 552, 23, 131072))
    g = capi.get_params()
    for k, _ in g._fields_:
        assert getattr(g, k) == getattr(q, k), k
    assert (q.modLen, q.nttLen, q.logCrtPrime, q.numCrtPrime, q.numEvalKey) == (65536, 65536, 23, 48, 69)
    pr = O.gen_crt_primes(q)
    assert all(int(p) < (1 << 23) for p in pr) and 2 * 65536 * (int(max(pr)) - 1) ** 2 < O.P
    capi.lib.cuhe_hip_reset_parameters()


def test_gmp_kronecker_baseline_equals_the_oracle_path():
    """the optional GMP baseline of bench.py (one mpz_mul of Kronecker-packed operands, the shape of the NTL call the
    reference makes at examples/DHS/DHS.cu:219-221) against the oracle's own multiply; threads do not change results"""
    if not O.lib().orc_gmp_available():
        pytest.skip("no libgmp on this box")
    o = O.Ctx(3, 2, 16, 50, 25, 16384)
    try:
        q = o.prm
        for lvl in (0, 2):
            M, W = o.coeff_modulus(lvl), o.words(lvl)
            a, _ = O.random_raw(q.rawLen, q.modLen, W, M, 31 + lvl); b, _ = O.random_raw(q.rawLen, q.modLen, W, M, 41 + lvl)
            want = o.mul_raw(a, b, lvl)
            assert np.array_equal(O.gmp_mul_xn1(a, b, M), want)
            assert O.set_threads(0) >= 1
            assert np.array_equal(o.mul_raw(a, b, lvl), want)
            O.set_threads(1)
    finally:
        O.set_threads(1); o.close()


def test_batched_chain_equals_the_cyclic_chain_of_the_oracle():
    """orc_nc_mul_relin_crt_batch (cAnd + relin per prime through the negacyclic restatement, the form that fits BASELINE
    config 4) against orc_mul_relin_crt (the reference-shaped chain: zero-padded cyclic transforms, exact remainder,
    np x K x nttLen key table; cuhe/CuHE.cu:101,570-581) on x^8192 + 1 with dense random keys, two levels, three pairs."""
    args = (3, 2, 16, 50, 25, 16384)
    o = O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xD100 + j)[0] for j in range(K)])
        ek, ekc = o.init_relin(ek_raw), o.key_residues(ek_raw)
        assert ekc.shape == (K, q.numCrtPrime, q.crtLen)
        for lvl in (0, 1):
            npr = o.np_(lvl)
            xs = [o.crt(O.random_raw(q.rawLen, q.modLen, o.words(lvl), o.coeff_modulus(lvl), 0xD200 + 16 * lvl + i)[0], lvl) for i in range(6)]
            a, b = np.stack(xs[0::2]), np.stack(xs[1::2])
            got = o.nc_mul_relin_crt_batch(a, b, lvl, ekc)
            for t in range(3):
                assert np.array_equal(got[t], o.mul_relin_crt(a[t], b[t], lvl, ek)), (lvl, t)
            # the prepared form bench.py times as the CPU leg of mul + relin (keys transformed beforehand, shared tables), 1 and 4 threads
            for threads in (1, 4):
                O.set_threads(threads)
                h = o.nc_prepare(lvl, ekc)
                try:
                    for t in range(3):
                        assert np.array_equal(o.nc_mul_relin_prepared(h, a[t], b[t], lvl), got[t]), (lvl, t, threads)
                finally:
                    o.nc_prepared_free(h)
                    O.set_threads(1)
    finally:
        o.close()
    # a ring that is not x^n + 1 is refused
    o = O.Ctx(3, 2, 8, 40, 20, 1155)
    try:
        z = np.zeros((1, o.np_(0), o.prm.crtLen), dtype=np.uint32)
        out = np.empty_like(z)
        ekc = np.zeros((o.prm.numEvalKey, o.prm.numCrtPrime, o.prm.crtLen), dtype=np.uint32)
        assert O.lib().orc_nc_mul_relin_crt_batch(o.h, O._p(out), O._p(z), O._p(z), 1, 0, O._p(ekc)) == -1
    finally:
        o.close()
