"""-m gpu parity tests: every HIP stage, called through the C ABI, against the
oracle on the same seeded inputs (bit-exact), plus the committed golden vectors.
Mirrors the reference's own tests: tests/test_ModP.cu (field ops vs big-int),
tests/test_ntt.cu:38-64 (transform vs definition), and the pipeline identities
of examples/DHS/DHS.cu:219-221."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X (no CPU fallback exists for the HIP path)")
    import gpu_util
    return gpu_util


def test_native_library_loaded(gu):
    assert b"gfx950" in gu.lib.cuhe_hip_version()
    maps = open("/proc/self/maps").read()
    assert "libcuhe_hip.so" in maps


def test_modp_ops(gu):
    """tests/test_ModP.cu:38-135: add/sub/mul on 2^20 random 64-bit inputs + shifts 3ab."""
    import oracle_lib as O
    rng = np.random.default_rng(1)
    n = 1 << 20
    x = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    y = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    edge = np.array([0, 1, O.P - 1, O.P, O.P + 1, (1 << 64) - 1, 0xFFFFFFFF, 1 << 32, (1 << 32) - 1], dtype=np.uint64)
    x[:81] = np.repeat(edge, 9); y[:81] = np.tile(edge, 9)
    # borrow with the low word of the difference all ones (the fix-up's carry chain crosses the word boundary: the path a
    # random pair takes with probability 2^-32), and the neighbours of it
    for i in range(1024):
        hi_a, hi_b = int(rng.integers(0, 0xFFFFFFFF)), int(rng.integers(0, 0xFFFFFFFF))
        if hi_a > hi_b:
            hi_a, hi_b = hi_b, hi_a
        lo_a = int(rng.integers(0, 0xFFFFFFFF)); delta = (-1, 0, 1, 2)[i & 3]
        x[100 + i] = (hi_a << 32) | lo_a
        y[100 + i] = (hi_b << 32) | ((lo_a + 1 + delta) & 0xFFFFFFFF)
    dx, dy, dz = gu.to_dev(x), gu.to_dev(y), gu.empty_u64(n)
    xi = [int(v) for v in x[:4096]]; yi = [int(v) for v in y[:4096]]
    P = O.P
    for name, f in (("add", lambda a, b: (a + b) % P), ("sub", lambda a, b: (a - b) % P), ("mul", lambda a, b: a * b % P)):
        gu.ck(getattr(gu.lib, "cuhe_hip_modp_" + name)(dz.data_ptr(), dx.data_ptr(), dy.data_ptr(), n, 0, None))
        got = gu.host_u64(dz)
        assert [int(v) for v in got[:4096]] == [f(a, b) for a, b in zip(xi, yi)], name
        # whole array against vectorised numpy big-int emulation through python ints on a stride
        for i in range(4096, n, 997):
            assert int(got[i]) == f(int(x[i]), int(y[i])), (name, i)
    for a in range(8):
        for b in range(8):
            l = 3 * a * b
            gu.ck(gu.lib.cuhe_hip_modp_shl(dz.data_ptr(), dx.data_ptr(), l, 4096, 0, None))
            got = gu.host_u64(dz)[:4096]
            assert [int(v) for v in got] == [(v << l) % P for v in xi], l


@pytest.mark.parametrize("length", [16384, 32768, 65536])
def test_ntt_fwd_batched_vs_oracle_and_golden(gu, golden, length):
    import oracle_lib as O
    g = golden("ntt.json")[str(length)]
    batch = 11                                   # odd: exercises the XCD-mapping guard
    xs = [O.splitmix_u32_below(length // 2, g["bound"], g["seed"])]
    xs.append(O.splitmix_u32_below(length // 2, 0xFFFFFFFF, g["seed32"]))
    xs.append(np.zeros(length // 2, dtype=np.uint32))
    xs.append(np.full(length // 2, 0xFFFFFFFF, dtype=np.uint32))
    one = np.zeros(length // 2, dtype=np.uint32); one[0] = 1
    xs.append(one)
    while len(xs) < batch:
        xs.append(O.splitmix_u32_below(length // 2, 0xFFFFFFFF, 1000 + len(xs)))
    x = np.stack(xs)
    dx, dX = gu.to_dev(x), gu.empty_u64(batch, length)
    gu.ck(gu.lib.cuhe_hip_ntt_prepare(length, 0))
    gu.ck(gu.lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), dx.data_ptr(), length, batch, length // 2, 0, None))
    X = gu.host_u64(dX)
    assert sha(X[0]) == g["sha256_full"]
    assert sha(X[1]) == g["sha256_full32"]
    for i, v in g["by_definition"].items():      # tests/test_ntt.cu:44-55
        assert int(X[0][int(i)]) == int(v)
    assert not X[2].any() and (X[4] == 1).all()
    for b in range(batch):
        assert np.array_equal(X[b], O.ntt_ext(x[b], length)), b


def test_ntt_chunking(gu):
    """more transforms than one scratch slab: the driver splits the batch."""
    import oracle_lib as O
    length, batch = 16384, 37
    gu.ck(gu.lib.cuhe_hip_set_ntt_chunk(16))
    try:
        x = np.stack([O.splitmix_u32_below(length // 2, 0xFFFFFFFF, 50 + b) for b in range(batch)])
        dx, dX = gu.to_dev(x), gu.empty_u64(batch, length)
        gu.ck(gu.lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), dx.data_ptr(), length, batch, length // 2, 0, None))
        X = gu.host_u64(dX)
        for b in range(batch):
            assert np.array_equal(X[b], O.ntt_ext(x[b], length)), b
    finally:
        gu.ck(gu.lib.cuhe_hip_set_ntt_chunk(0))


PSETS = {
    "toy1155": (3, 2, 8, 40, 20, 1155),          # composite m: generic NTT Barrett
    "dhs_simple": (5, 2, 1, 61, 20, 8191),       # prime m: fold reduction
    "pow2_16384": (3, 2, 16, 50, 25, 16384),     # x^n + 1
    "prince_small": (3, 2, 16, 25, 25, 21845),   # Prince ring (n=16384, L=32768), 3 levels
    "pow2_32768": (3, 2, 16, 48, 24, 32768),     # x^16384 + 1 with 24-bit primes: the NEGACYCLIC ciphertext domain applies
    # general Phi_m behind 64K-point transforms: what the reference's ntt_*_64k kernels exist for (cuhe/Operations.cu:322-329,
    # 460-501, cuhe/Base.cu:659-842,927-1001).  Depth 3 / 4 primes: the oracle's Barrett chain at L = 65536 takes ~2 s per call.
    "phi32767": (3, 2, 16, 50, 25, 32767),       # m = 7 * 31 * 151, phi = 27000: generic (folded / five-transform) Barrett, L = 65536
    "prime32749": (3, 2, 16, 50, 25, 32749),     # prime m, n = 32748: the prime-m fold at L = 65536
    "prime16381": (3, 2, 16, 50, 25, 16381),     # prime m, n = 16380: the prime-m fold at L = 32768
    "phi65535": (3, 2, 16, 50, 25, 65535),       # m = 3 * 5 * 17 * 257, phi = 32768 = the whole row: the quotient (32767 coefficients) does not fit the
}                                                # folded form, so the DEFAULT reduction is the five-transform chain at L = 65536 (Operations.cu:460-501)
GENERAL_64K = ["phi32767", "prime32749", "prime16381", "phi65535"]


@pytest.fixture(scope="module", params=list(PSETS))
def ctxpair(request, gu):
    import oracle_lib as O
    args = PSETS[request.param]
    g = gu.GpuCtx(*args)
    o = O.Ctx(*args)
    yield request.param, g, o
    g.close(); o.close()


def _rand_crt(o, np_, seed, full=False):
    q = o.prm
    rng = np.random.default_rng(seed)
    a = np.zeros((np_, q.crtLen), dtype=np.uint32)
    for i in range(np_):
        a[i, :q.modLen] = (o.primes[i] - 1) if full else rng.integers(0, o.primes[i], q.modLen)
    return a


def test_ctx_constants(ctxpair):
    name, g, o = ctxpair
    assert np.array_equal(g.primes, o.primes)
    for k, _ in g.prm._fields_:
        assert getattr(g.prm, k) == getattr(o.prm, k), k
    for lvl in range(o.prm.depth):
        assert g.coeff_modulus(lvl) == o.coeff_modulus(lvl)
        assert g.words(lvl) == o.words(lvl) and g.np_(lvl) == o.np_(lvl)


def test_second_init_on_the_same_ring_keeps_the_context(gu):
    """cuhe_hip_init on the ring the library already runs on is a no-op that keeps tables, resident keys and blocks handed out
    (cuhe_hip_same_ring) -- what a second scheme object built from a key string does to the process (examples/DHS/DHS.cu:57-118,
    simple_DHS.cu:176-190); another ring is refused until cuhe_hip_shutdown."""
    import ctypes as C
    import oracle_lib as O
    lib, ck = gu.lib, gu.ck
    args = (3, 2, 8, 40, 20, 1155)
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE700 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        g.init_relin(ek_raw)
        ct, _ = O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE7FF)
        want = o.relin(ct, 0, ek)
        blk = lib.cuhe_hip_malloc(0, 4096)
        gen = lib.cuhe_hip_generation()
        assert lib.cuhe_hip_same_ring(None, 0) == 1
        phi = np.ascontiguousarray(O.cyclotomic(q.mSize), dtype=np.int32)
        assert lib.cuhe_hip_same_ring(phi.ctypes.data_as(C.c_void_p), phi.size) == 1
        ck(lib.cuhe_hip_set_parameters(*args))                       # the second object's setParameters ...
        ck(lib.cuhe_hip_init(None, 0))                               # ... and initCuHE
        ck(lib.cuhe_hip_init(phi.ctypes.data_as(C.c_void_p), phi.size))
        assert lib.cuhe_hip_generation() == gen and lib.cuhe_hip_is_initialised() == 1
        assert np.array_equal(g.relin(ct, 0), want)                  # the keys loaded before are still there
        ck(lib.cuhe_hip_free(0, blk))                                # and so is the block handed out before
        other = phi.copy(); other[1] += 1                            # another modulus on the same parameters: not the same ring
        assert lib.cuhe_hip_same_ring(other.ctypes.data_as(C.c_void_p), other.size) == 0
        assert lib.cuhe_hip_init(other.ctypes.data_as(C.c_void_p), other.size) != 0 and b"another ring" in lib.cuhe_hip_last_error()
        ck(lib.cuhe_hip_set_parameters(3, 2, 8, 40, 20, 8191))       # other parameters
        assert lib.cuhe_hip_same_ring(None, 0) == 0 and lib.cuhe_hip_init(None, 0) != 0
        ck(lib.cuhe_hip_set_parameters(*args))
        assert lib.cuhe_hip_same_ring(None, 0) == 1 and np.array_equal(g.relin(ct, 0), want)
    finally:
        g.close(); o.close()


def test_crt_icrt(ctxpair):
    import oracle_lib as O
    name, g, o = ctxpair
    q = o.prm
    for lvl in (0, q.depth - 1):
        W, M = o.words(lvl), o.coeff_modulus(lvl)
        raw, vals = O.random_raw(q.rawLen, q.modLen, W, M, 77 + lvl)
        # edge coefficients: 0, 1, M-1, and the largest W-word value (unreduced input)
        vals[0], vals[1], vals[2] = 0, 1, M - 1
        raw = O.ints_to_raw(vals, q.rawLen, W)
        raw[3] = 0xFFFFFFFF
        crt_g = g.crt(raw, lvl)
        assert np.array_equal(crt_g, o.crt(raw, lvl))
        back = g.icrt(crt_g, lvl)
        assert np.array_equal(back, o.icrt(crt_g, lvl))
        raw[3] = 0
        assert np.array_equal(g.icrt(g.crt(raw, lvl), lvl), raw)


def test_crt_icrt_kernel_forms_equal_each_other_and_oracle(ctxpair, gu):
    """cuhe_hip_set_icrt_mfma: the column sums on the matrix cores (icrt_mfma.cuh) against the VALU kernel and the oracle, at
    EVERY level: reduced rows of edge values (0, small, M - small, multiples of M / np: quotient estimates next to an integer)
    and unreduced rows (x_i up to 2^32 - 1)."""
    import oracle_lib as O
    name, g, o = ctxpair
    q = o.prm
    try:
        for lvl in range(q.depth):
            W, M, npl = o.words(lvl), o.coeff_modulus(lvl), o.np_(lvl)
            raw, vals = O.random_raw(q.rawLen, q.modLen, W, M, 177 + lvl)
            edge = [0, 1, 2, M - 1, M - 2, M // 2, M // 2 + 1] + [k * M // npl + e for k in range(1, min(npl, 6)) for e in (-1, 0, 1)]
            for i, v in enumerate(edge[:q.modLen]):
                vals[i] = v % M
            raw = O.ints_to_raw(vals, q.rawLen, W)
            rows = o.crt(raw, lvl)
            rng = np.random.default_rng(500 + lvl)
            wild = rows.copy()
            wild[:, :q.modLen] = rng.integers(0, 1 << 32, (npl, q.modLen), dtype=np.uint64).astype(np.uint32)
            wild[:, :4] = 0xFFFFFFFF
            for on in (1, 0):                                    # the CRT with 64-bit sums and with the carry word (cuhe_hip_set_crt_acc64)
                gu.ck(gu.lib.cuhe_hip_set_crt_acc64(on))
                assert np.array_equal(g.crt(raw, lvl), rows), (name, lvl, on)
                full = np.full_like(raw, 0xFFFFFFFF)              # every word 2^32 - 1: the largest sums
                assert np.array_equal(g.crt(full, lvl), o.crt(full, lvl)), (name, lvl, on)
            out = {}
            for on in (1, 0):
                gu.ck(gu.lib.cuhe_hip_set_icrt_mfma(on))
                out[on] = (g.icrt(rows, lvl), g.icrt(wild, lvl))
            assert np.array_equal(out[1][0], raw), (name, lvl)
            assert np.array_equal(out[0][0], raw), (name, lvl)
            assert np.array_equal(out[1][1], out[0][1]), (name, lvl)
            assert np.array_equal(out[1][1], o.icrt(wild, lvl)), (name, lvl)
        assert gu.lib.cuhe_hip_set_icrt_mfma(2) != 0
        assert gu.lib.cuhe_hip_set_crt_acc64(2) != 0
    finally:
        gu.lib.cuhe_hip_set_icrt_mfma(1)
        gu.lib.cuhe_hip_set_crt_acc64(1)


def test_ntt_intt_roundtrip_and_oracle(ctxpair):
    name, g, o = ctxpair
    q = o.prm
    for lvl in (0, q.depth - 1):
        np_ = o.np_(lvl)
        a = _rand_crt(o, np_, 5 + lvl)
        X = g.ntt(a, lvl)
        assert np.array_equal(X, o.ntt(a))
        assert np.array_equal(g.intt(X, lvl), a)                 # BASELINE config 2: identity check
        assert np.array_equal(g.intt_double_deg(X, lvl), o.intt_hold(X))


def test_pointwise(ctxpair):
    name, g, o = ctxpair
    q = o.prm
    np_ = o.np_(0)
    rng = np.random.default_rng(9)
    import oracle_lib as O
    x = rng.integers(0, O.P, (np_, q.nttLen), dtype=np.uint64)
    y = rng.integers(0, O.P, (np_, q.nttLen), dtype=np.uint64)
    x[0, :4] = [0, 1, O.P - 1, O.P - 1]; y[0, :4] = [O.P - 1, O.P - 1, O.P - 1, 1]
    s = y[0].copy()
    assert np.array_equal(g.ntt_mul(x, y, 0), o.ntt_mul(x, y))
    assert np.array_equal(g.ntt_add(x, y, 0), o.ntt_add(x, y))
    assert np.array_equal(g.ntt_mul_nx1(x, s, 0), o.ntt_mul_nx1(x, s))
    assert np.array_equal(g.ntt_add_nx1(x, s, 0), o.ntt_add_nx1(x, s))
    a, b = _rand_crt(o, np_, 1), _rand_crt(o, np_, 2, full=True)
    assert np.array_equal(g.crt_add(a, b, 0), o.crt_add(a, b))
    assert np.array_equal(g.crt_add_int(a, q.modMsg - 1, 0), o.crt_add_int(a, q.modMsg - 1))   # cNot (CuHE.cu:197)
    assert np.array_equal(g.crt_add_int(a, 0xFFFFFFF0, 0), o.crt_add_int(a, 0xFFFFFFF0))
    assert np.array_equal(g.crt_add_nx1(a, b[0], 0), o.crt_add_nx1(a, b[0]))


def test_intt_mod_both_paths(ctxpair, gu):
    name, g, o = ctxpair
    q = o.prm
    for lvl in (0, q.depth - 1):
        np_ = o.np_(lvl)
        for full in (False, True):
            a, b = _rand_crt(o, np_, 11, full), _rand_crt(o, np_, 12, full)
            X = o.ntt_mul(o.ntt(a), o.ntt(b))
            want = o.intt_mod(X)
            assert np.array_equal(g.intt_mod(X, lvl), want)
            for force in (1, 2):       # 1: generic path (folded form where the ring allows it); 2: the five-transform
                gu.ck(gu.lib.cuhe_hip_force_generic_reduce(force))       # form of cuhe/Operations.cu:460-501
                try:
                    assert gu.lib.cuhe_hip_reduce_kind() == 0
                    assert np.array_equal(g.intt_mod(X, lvl), want)
                    assert np.array_equal(g.barrett(o.intt_hold(X), lvl), want)
                finally:
                    gu.ck(gu.lib.cuhe_hip_force_generic_reduce(0))


def test_modswitch(ctxpair):
    name, g, o = ctxpair
    q = o.prm
    for lvl in range(q.depth - 1):
        a = _rand_crt(o, o.np_(lvl), 21 + lvl)
        a[:, 0] = 0; a[-1, 1] = o.primes[o.np_(lvl) - 1] - 1; a[-1, 2] = 1
        want = o.modswitch(a)
        assert np.array_equal(g.modswitch(a, lvl), want)
        assert np.array_equal(g.modswitch(a, lvl, in_place=True), want)


def test_mul_pipeline_vs_oracle(ctxpair):
    import oracle_lib as O
    name, g, o = ctxpair
    q = o.prm
    for lvl in (0, q.depth - 1):
        W, M = o.words(lvl), o.coeff_modulus(lvl)
        a, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xA000 + 17 * lvl)
        b, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xB000 + 31 * lvl)
        assert np.array_equal(g.mul_raw(a, b, lvl), o.mul_raw(a, b, lvl))


@pytest.mark.parametrize("name,fixture", [("toy1155", "pipeline_toy1155.json"),
                                          ("pow2_16384", "pipeline_pow2_16384.json"),
                                          ("dhs_simple", "pipeline_dhs_simple.json"),
                                          ("c1_pow2_1prime", "pipeline_c1_pow2_1prime.json"),         # BASELINE config 1: N = 2^13,
                                          ("c1_prime_m_1prime", "pipeline_c1_prime_m_1prime.json")])  # exactly ONE CRT prime
def test_mul_pipeline_vs_golden(gu, golden, name, fixture):
    """(a*b mod Phi_m) mod q against the pure-Python big-int fixtures (examples/DHS/DHS.cu:219-221)."""
    import oracle_lib as O
    gold = golden(fixture)
    g = gu.GpuCtx(*gold["args"])
    try:
        q = g.prm
        if name == "c1_pow2_1prime":                   # x^8192 + 1: the ciphertext domain is the negacyclic transform of 8192 points
            assert g.nc and g.ctlen == 8192 == q.modLen
        for lvl_s, rec in gold["levels"].items():
            lvl = int(lvl_s)
            W, M = g.words(lvl), g.coeff_modulus(lvl)
            a, _ = O.random_raw(q.rawLen, q.modLen, W, M, rec["seed_a"])
            b, _ = O.random_raw(q.rawLen, q.modLen, W, M, rec["seed_b"])
            crt_a = g.crt(a, lvl)
            assert sha(crt_a) == rec["crt_a_sha256"]
            assert sha(g.mul_raw(a, b, lvl)) == rec["mul_sha256"]
            assert sha(g.crt_add(crt_a, g.crt(b, lvl), lvl)) == rec["crt_add_sha256"]
            if "modswitch_sha256" in rec:
                assert sha(g.modswitch(crt_a, lvl)) == rec["modswitch_sha256"]
        if "relin" in gold:
            K, W0, M0 = q.numEvalKey, g.words(0), g.coeff_modulus(0)
            ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, gold["relin"]["seed_ek_base"] + j)[0]
                               for j in range(K)])
            g.init_relin(ek_raw)
            for lvl_s, rec in gold["relin"]["levels"].items():
                lvl = int(lvl_s)
                ct, _ = O.random_raw(q.rawLen, q.modLen, g.words(lvl), g.coeff_modulus(lvl), rec["seed_ct"])
                res = g.relin_crt(ct, lvl)
                assert sha(res) == rec["crt_sha256"]
    finally:
        g.close()


@pytest.mark.parametrize("name", ["prince_small"] + GENERAL_64K)
def test_relin_vs_oracle(gu, name):
    """window NTTs + key-switch inner product + mul+relin chain (cuhe/CuHE.cu:570-581): the Prince ring (w = 16) and general Phi_m
    behind 64K-point transforms (composite m; prime m at 64K and 32K points)."""
    import oracle_lib as O
    args = PSETS[name]
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE000 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        g.init_relin(ek_raw)
        for lvl in (0, 1):
            ct, _ = O.random_raw(q.rawLen, q.modLen, o.words(lvl), o.coeff_modulus(lvl), 0xC100 + lvl)
            assert np.array_equal(g.nttw(ct, lvl), o.nttw(ct, lvl))
            assert np.array_equal(g.relin(ct, lvl), o.relin(ct, lvl, ek))
            np_ = o.np_(lvl)
            a, b = _rand_crt(o, np_, 31), _rand_crt(o, np_, 32)
            want = o.mul_relin_crt(a, b, lvl, ek)
            assert np.array_equal(g.mul_relin_crt(a, b, lvl), want)
            assert np.array_equal(g.mul_relin_crt(a, b, lvl, fused=True), want), lvl          # relinearization ; n2c as one call (round 5)
    finally:
        g.close(); o.close()


@pytest.mark.parametrize("name", ["toy1155", "pow2_16384", "pow2_32768"] + GENERAL_64K)
def test_fused_relin_chain_equals_the_two_calls(gu, name):
    """cuhe_hip_relin_crt (raw -> reduced CRT rows in one call, what CuCtxt::relin uses since round 5) against cuhe_hip_relinearization ;
    cuhe_hip_ct_intt on cyclic and negacyclic rings, every level, repeated (scratch reuse), on a caller's own stream too."""
    import ctypes as C
    import oracle_lib as O
    lib, ck = gu.lib, gu.ck
    g = gu.GpuCtx(*PSETS[name])
    try:
        q = g.prm
        K, W0, M0 = q.numEvalKey, g.words(0), g.coeff_modulus(0)
        g.init_relin(np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xF000 + j)[0] for j in range(K)]))
        st = C.c_void_p()
        ck(lib.cuhe_hip_stream_create(0, C.byref(st)))
        for lvl in range(q.depth):
            npr = g.np_(lvl)
            raw, _ = O.random_raw(q.rawLen, q.modLen, g.words(lvl), g.coeff_modulus(lvl), 0xF100 + lvl)
            d_raw = gu.to_dev(raw)
            acc = gu.empty_u64(npr, g.ctlen)
            want = gu.empty_u32(npr, q.crtLen); want.zero_()
            ck(lib.cuhe_hip_relinearization(acc.data_ptr(), d_raw.data_ptr(), lvl, 0, None))
            ck(lib.cuhe_hip_ct_intt(want.data_ptr(), acc.data_ptr(), g.logq(lvl), 1, 0, None))
            w = gu.host_u32(want)
            for rep in range(2):
                for stream in (None, st):
                    got = gu.empty_u32(npr, q.crtLen); got.zero_()
                    ck(lib.cuhe_hip_relin_crt(got.data_ptr(), d_raw.data_ptr(), lvl, 0, stream))
                    ck(lib.cuhe_hip_stream_sync(0, stream))
                    assert np.array_equal(gu.host_u32(got)[:, :q.modLen], w[:, :q.modLen]), (name, lvl, rep, stream is not None)
        assert lib.cuhe_hip_relin_crt(want.data_ptr(), d_raw.data_ptr(), q.depth, 0, None) != 0
        ck(lib.cuhe_hip_stream_destroy(0, st))
    finally:
        g.close()


@pytest.mark.parametrize("nc", [True, False], ids=["negacyclic", "cyclic"])
def test_config3_full_size_properties(gu, nc):
    """BASELINE config 3: N = 2^15 (L = 65536), 32 CRT primes, full ciphertext multiply, in the negacyclic ciphertext
    domain (32K-point transforms, the default on this ring) and in the reference's cyclic one (64K-point transforms).
    Size-independent properties: (1) CRT->ICRT and NTT->INTT round trips, (2) x * 1 = x,
    (3) x * x^k = negacyclic shift, (4) bilinearity (a+b)*c = a*c + b*c mod q, plus the oracle on
    one prime row."""
    import oracle_lib as O
    args = (9, 2, 16, 576, 24, 65536)
    g = gu.GpuCtx(*args, negacyclic=nc)
    try:
        q = g.prm
        assert q.nttLen == 65536 and q.numCrtPrime == 32 and q.modLen == 32768 and g.nc == nc
        lvl = 0
        W, M, n = g.words(lvl), g.coeff_modulus(lvl), q.modLen
        a, av = O.random_raw(q.rawLen, n, W, M, 1)
        b, bv = O.random_raw(q.rawLen, n, W, M, 2)
        c, cv = O.random_raw(q.rawLen, n, W, M, 3)
        assert np.array_equal(g.icrt(g.crt(a, lvl), lvl), a)
        ca = g.crt(a, lvl)
        X = g.ntt(ca, lvl)
        assert np.array_equal(g.intt(X, lvl), ca)
        assert np.array_equal(X[5], O.ntt_ext(ca[5], q.nttLen))
        cb = g.crt(b, lvl)
        pr = g.ct_intt(g.ct_mul(g.ct_ntt(ca, lvl), g.ct_ntt(cb, lvl), lvl), lvl, True)      # dense product rows vs the oracle
        for i in (0, 13, 31):
            assert np.array_equal(pr[i], O.negacyclic_mul_modp(ca[i], cb[i], int(g.primes[i]))), i
        one = O.ints_to_raw([1], q.rawLen, W)
        assert np.array_equal(g.mul_raw(a, one, lvl), a)
        k = 12345
        xk = O.ints_to_raw([0] * k + [1], q.rawLen, W)
        sh = g.mul_raw(a, xk, lvl)
        want = [0] * n
        for i in range(n):                        # x^n = -1
            j = i + k
            if j < n: want[j] = av[i]
            else: want[j - n] = (M - av[i]) % M
        assert O.raw_to_ints(sh, n) == want
        ab = O.ints_to_raw([(x + y) % M for x, y in zip(av, bv)], q.rawLen, W)
        lhs = O.raw_to_ints(g.mul_raw(ab, c, lvl), n)
        r1, r2 = O.raw_to_ints(g.mul_raw(a, c, lvl), n), O.raw_to_ints(g.mul_raw(b, c, lvl), n)
        assert lhs == [(x + y) % M for x, y in zip(r1, r2)]
    finally:
        g.close()


@pytest.mark.parametrize("nc", [True, False], ids=["negacyclic", "cyclic"])
def test_config3_full_size_vs_oracle(gu, nc):
    """BASELINE config 3 AT FULL SIZE against the oracle, every coefficient (this comparison used to live only inside
    bench.py): N = 2^15 (x^32768 + 1), 32 CRT primes, the full multiply raw -> raw = CRT -> NTT -> pointwise -> INTT -> ICRT
    (cuhe/CuHE.cu:259-268 mulZZX; examples/DHS/DHS.cu:219-221 is the host arithmetic it stands for), in the negacyclic ciphertext
    domain and in the reference's cyclic one (through the ciphertext-domain entry points AND through ntt / nttMul / inttMod), on
    reduced random operands, on unreduced ones (every word 32 random bits, what bench.py times), and for three operand pairs per call
    (cuhe_hip_mul_raw_batch).  The oracle runs with OpenMP over the primes (0.3 s per multiply on the GPU box's cores)."""
    import oracle_lib as O
    args = (9, 2, 16, 576, 24, 65536)
    g, o = gu.GpuCtx(*args, negacyclic=nc), O.Ctx(*args)
    used = O.set_threads(0)
    try:
        q = g.prm
        assert q.nttLen == 65536 and q.numCrtPrime == 32 and q.modLen == 32768 and g.nc == nc and used >= 1
        W, M = g.words(0), g.coeff_modulus(0)
        a, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xC3A0)
        b, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xC3B0)
        want = o.mul_raw(a, b, 0)
        assert np.array_equal(g.mul_raw(a, b, 0), want)
        if not nc:
            assert np.array_equal(g.mul_raw(a, b, 0, cyclic_api=True), want)
        rng = np.random.default_rng(0xC3)
        pairs = [(rng.integers(0, 1 << 32, (q.rawLen, W), dtype=np.uint64).astype(np.uint32),
                  rng.integers(0, 1 << 32, (q.rawLen, W), dtype=np.uint64).astype(np.uint32)) for _ in range(3)]
        wants = [o.mul_raw(x, y, 0) for x, y in pairs]
        assert np.array_equal(g.mul_raw(*pairs[0], 0), wants[0])
        da = gu.to_dev(np.concatenate([x for x, _ in pairs])); db = gu.to_dev(np.concatenate([y for _, y in pairs]))
        out = gu.empty_u32(3 * q.rawLen, W)
        gu.ck(gu.lib.cuhe_hip_mul_raw_batch(out.data_ptr(), da.data_ptr(), db.data_ptr(), 0, 3, 0, None))
        got = gu.host_u32(out)
        for i in range(3):
            assert np.array_equal(got[i * q.rawLen:(i + 1) * q.rawLen], wants[i]), i
        # one level down: 31 primes, one coefficient word less
        W1, M1 = g.words(1), g.coeff_modulus(1)
        a1, _ = O.random_raw(q.rawLen, q.modLen, W1, M1, 0xC3A1)
        b1, _ = O.random_raw(q.rawLen, q.modLen, W1, M1, 0xC3B1)
        assert np.array_equal(g.mul_raw(a1, b1, 1), o.mul_raw(a1, b1, 1))
    finally:
        O.set_threads(1)
        g.close(); o.close()


@pytest.mark.parametrize("name", ["toy1155", "prince_small", "pow2_32768", "phi32767", "prime32749"])
def test_prime_range_entry_points(gu, name):
    """CRT-prime-sharded entry points (cuhe_hip_*_range / *_rows): two shards computed one after the other on one
    GPU and reassembled must equal the unsharded oracle result (the collective itself is covered by
    tests/test_sharded_gloo.py)."""
    import torch
    import oracle_lib as O
    from cuhe_amd.sharded import HipBackend, ShardedMulRelin, shard_bounds
    args = PSETS[name]
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE000 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        g.init_relin(ek_raw)
        hb = HipBackend()
        for lvl in (0, 1):
            npr = o.np_(lvl)
            a, b = _rand_crt(o, npr, 41), _rand_crt(o, npr, 42)
            want = o.mul_relin_crt(a, b, lvl, ek)
            # crt_range on raw input
            W, M = o.words(lvl), o.coeff_modulus(lvl)
            raw, _ = O.random_raw(q.rawLen, q.modLen, W, M, 99 + lvl)
            crt_full = o.crt(raw, lvl)
            world = 2
            na = hb.ntt_rows(gu.to_dev(a)); nb = hb.ntt_rows(gu.to_dev(b))
            if not g.nc: assert np.array_equal(gu.host_u64(na), o.ntt(a))
            # stage 1 per shard, then emulate the all-gather by concatenation
            crt_rows = []
            for r in range(world):
                f, c = shard_bounds(npr, world, r)
                prod = hb.ntt_mul_rows(na[f:f + c].contiguous(), nb[f:f + c].contiguous())
                crt_rows.append(hb.intt_mod_range(prod, lvl, f, c))
                d = gu.empty_u32(c, q.crtLen)
                gu.ck(gu.lib.cuhe_hip_crt_range(d.data_ptr(), gu.to_dev(raw).data_ptr(), o.logq(lvl), f, c, 0, None))
                assert np.array_equal(gu.host_u32(d), crt_full[f:f + c])
            crt_all = torch.cat(crt_rows)
            rawp = hb.icrt(crt_all, lvl)
            outs = []
            for r in range(world):
                f, c = shard_bounds(npr, world, r)
                outs.append(hb.intt_mod_range(hb.relin_range(rawp, lvl, f, c), lvl, f, c))
            assert np.array_equal(gu.host_u32(torch.cat(outs)), want)
            # and the driver object with world == 1
            sh = ShardedMulRelin(hb, lvl, 0, 1)
            assert np.array_equal(gu.host_u32(sh.mul_relin(na, nb)), want)
    finally:
        g.close(); o.close()


def test_remaining_entry_points_and_ragged_cases(gu):
    """single-polynomial drivers (_ntt/_nttw/_intt, cuhe/Operations.h:78-81), inttHold + inttResult + barrett(dst,lvl),
    crtMulInt, an inverse batch with a prime offset and a store count that is not a multiple of the tile, and empty batches."""
    import ctypes as C
    import oracle_lib as O
    args = PSETS["dhs_simple"]                   # modLen = 8190 < crtLen = 8192: ragged rows
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q, lib, ck = o.prm, gu.lib, gu.ck
        L, cl = q.nttLen, q.crtLen
        npr = q.numCrtPrime
        a = _rand_crt(o, npr, 77)
        assert not a[:, q.modLen:].any()
        # _ntt on one row, _intt with an explicit prime index
        dx, dX = gu.to_dev(a[3]), gu.empty_u64(L)
        ck(lib.cuhe_hip_ntt_one(dX.data_ptr(), dx.data_ptr(), 0, None))
        X3 = gu.host_u64(dX)
        assert np.array_equal(X3, O.ntt_ext(a[3], L))
        dback = gu.empty_u32(L)
        ck(lib.cuhe_hip_intt_one(dback.data_ptr(), dX.data_ptr(), 3, 0, None))
        assert np.array_equal(gu.host_u32(dback), O.intt_modp(X3, L, int(o.primes[3])))
        # _nttw: window 5 of a raw polynomial
        W = o.words(0)
        raw, _ = O.random_raw(q.rawLen, q.modLen, W, o.coeff_modulus(0), 5)
        dW = gu.empty_u64(L)
        ck(lib.cuhe_hip_nttw_one(dW.data_ptr(), gu.to_dev(raw).data_ptr(), W, 5, 0, None))
        assert np.array_equal(gu.host_u64(dW), o.nttw(raw, 0)[5])
        # inverse batch: rows 2..5 reduced modulo primes 2..5, only the first 5000 outputs stored
        X = o.ntt(a)
        nst = 5000
        d = gu.empty_u32(4, L)
        ck(lib.cuhe_hip_ntt_inv_batched(d.data_ptr(), gu.to_dev(X[2:6]).data_ptr(), L, 4, L, nst, 2, 0, None))
        got = gu.host_u32(d)
        for r in range(4):
            want = O.intt_modp(X[2 + r], L, int(o.primes[2 + r]))
            assert np.array_equal(got[r, :nst], want[:nst]) and not got[r, nst:].any()
        # empty batches are no-ops
        ck(lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), dx.data_ptr(), L, 0, L // 2, 0, None))
        ck(lib.cuhe_hip_ntt_inv_batched(d.data_ptr(), dX.data_ptr(), L, 0, L, L, 0, 0, None))
        assert lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), dx.data_ptr(), 12345, 1, 8192, 0, None) != 0    # bad length
        # inttHold -> inttResult -> barrett(dst, lvl)
        b = _rand_crt(o, npr, 78)
        P = o.ntt_mul(X, o.ntt(b))
        ck(lib.cuhe_hip_intt_hold(gu.to_dev(P).data_ptr(), o.logq(0), 0, None))
        hold = lib.cuhe_hip_intt_result(0)
        assert hold
        dr = gu.empty_u32(npr, cl)
        ck(lib.cuhe_hip_barrett_hold(dr.data_ptr(), 0, 0, None))
        assert np.array_equal(gu.host_u32(dr), o.intt_mod(P))
        # crtMulInt: constant term times an integer (cuhe/Base.cu:1078-1087)
        da = gu.to_dev(a)
        ck(lib.cuhe_hip_crt_mul_int(da.data_ptr(), da.data_ptr(), 12345, o.logq(0), 0, None))
        got = gu.host_u32(da)
        want = a.copy()
        want[:, 0] = (a[:, 0].astype(np.uint64) * 12345 % o.primes.astype(np.uint64)).astype(np.uint32)
        assert np.array_equal(got, want)
    finally:
        g.close(); o.close()


def test_prince_full_size_properties(gu):
    """BASELINE config 5 parameters (examples/Prince/Prince.cu:48-49): n = 16384, L = 32768, 25 primes, composite
    m = 21845, generic NTT Barrett.  Size-independent properties of exact arithmetic in Z_q[x]/Phi_m:
    x^m = 1, so (a * x^k) * x^(m-k) == a;  a * 1 == a;  bilinearity;  CRT->ICRT round trip."""
    import oracle_lib as O
    g = gu.GpuCtx(25, 2, 16, 25, 25, 21845)
    try:
        q = g.prm
        assert (q.modLen, q.nttLen, q.numCrtPrime, q.numEvalKey) == (16384, 32768, 25, 40)
        assert gu.lib.cuhe_hip_reduce_kind() == 0
        m, n = q.mSize, q.modLen
        for lvl in (0, 24):
            W, M = g.words(lvl), g.coeff_modulus(lvl)
            a, av = O.random_raw(q.rawLen, n, W, M, 11 + lvl)
            b, bv = O.random_raw(q.rawLen, n, W, M, 12 + lvl)
            c, cv = O.random_raw(q.rawLen, n, W, M, 13 + lvl)
            assert np.array_equal(g.icrt(g.crt(a, lvl), lvl), a)
            assert np.array_equal(g.mul_raw(a, O.ints_to_raw([1], q.rawLen, W), lvl), a)
            k = 10000
            xk = O.ints_to_raw([0] * k + [1], q.rawLen, W)
            xmk = O.ints_to_raw([0] * (m - k) + [1], q.rawLen, W)
            assert np.array_equal(g.mul_raw(g.mul_raw(a, xk, lvl), xmk, lvl), a)
            ab = O.ints_to_raw([(x + y) % M for x, y in zip(av, bv)], q.rawLen, W)
            lhs = O.raw_to_ints(g.mul_raw(ab, c, lvl), n)
            r1, r2 = O.raw_to_ints(g.mul_raw(a, c, lvl), n), O.raw_to_ints(g.mul_raw(b, c, lvl), n)
            assert lhs == [(x + y) % M for x, y in zip(r1, r2)]
    finally:
        g.close()


def test_evalkey_cache_roundtrip(gu):
    """binary evaluation-key cache (include/cuhe_hip.h: cuhe_hip_relin_export / _import): an exported image, re-imported
    into a FRESH context of the same parameters, relinearises exactly like the context that computed the keys and
    like the oracle; images for other parameters, truncated images and damaged payloads are refused."""
    import ctypes
    import oracle_lib as O
    args = (3, 2, 16, 25, 25, 21845)
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE000 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        assert gu.lib.cuhe_hip_relin_export(None, 0, 0) != 0          # nothing to export before initRelinearization
        g.init_relin(ek_raw)
        size = gu.lib.cuhe_hip_relin_cache_size()
        assert size == 96 + q.numCrtPrime * K * q.nttLen * 8
        img = np.zeros(size, dtype=np.uint8)
        assert gu.lib.cuhe_hip_relin_export(img.ctypes.data_as(ctypes.c_void_p), size - 1, 0) != 0   # too small
        gu.ck(gu.lib.cuhe_hip_relin_export(img.ctypes.data_as(ctypes.c_void_p), size, 0))
        assert bytes(img[:8]) == b"CUHEEK\x00\x01"
        keys = img[96:].view(np.uint64).reshape(q.numCrtPrime, K, q.nttLen)
        assert np.array_equal(keys, np.asarray(ek).reshape(q.numCrtPrime, K, q.nttLen))    # payload = oracle's NTT-domain keys
        ct, _ = O.random_raw(q.rawLen, q.modLen, o.words(1), o.coeff_modulus(1), 0xC1)
        want = o.relin(ct, 1, ek)
        assert np.array_equal(g.relin(ct, 1), want)
    finally:
        g.close()
    g = gu.GpuCtx(*args)                                                # fresh context: no keys yet
    try:
        bad = img.copy(); bad[96 + 12345] ^= 1
        assert gu.lib.cuhe_hip_relin_import(bad.ctypes.data_as(ctypes.c_void_p), size) != 0           # damaged payload
        assert gu.lib.cuhe_hip_relin_import(img.ctypes.data_as(ctypes.c_void_p), size - 8) != 0       # truncated
        bad = img.copy(); w = bad[96:].view(np.uint64); w[[10, 11]] = w[[11, 10]]                      # two payload words swapped
        assert not np.array_equal(bad, img) and gu.lib.cuhe_hip_relin_import(bad.ctypes.data_as(ctypes.c_void_p), size) != 0
        bad = img.copy(); w = bad[96:].view(np.uint64); w[5] ^= np.uint64(1 << 40); w[9] ^= np.uint64(1 << 40)   # paired bit flips
        assert gu.lib.cuhe_hip_relin_import(bad.ctypes.data_as(ctypes.c_void_p), size) != 0
        bad = img.copy(); bad[0] = ord("X")
        assert gu.lib.cuhe_hip_relin_import(bad.ctypes.data_as(ctypes.c_void_p), size) != 0           # bad magic
        gu.ck(gu.lib.cuhe_hip_relin_import(img.ctypes.data_as(ctypes.c_void_p), size))
        assert np.array_equal(g.relin(ct, 1), want)
    finally:
        g.close()
    g = gu.GpuCtx(3, 2, 8, 25, 25, 21845)                              # other window size: the image must be refused
    try:
        assert gu.lib.cuhe_hip_relin_import(img.ctypes.data_as(ctypes.c_void_p), size) != 0
    finally:
        g.close(); o.close()


def test_reentrant_host_threads(gu):
    """Several host threads drive ONE GPU through the C ABI at the same time, each on its own stream
    (cuhe_hip_stream_create): the library keeps its scratch per host thread (the reference has one set per device and is
    not re-entrant, cuhe/Operations.cu:171-209).  Every thread runs the mul + relin chain (generic Barrett reduction,
    window transforms, key-switch inner product) on its own operands; all results must equal the oracle's."""
    import ctypes
    import threading
    import torch
    import oracle_lib as O
    args = (3, 2, 8, 40, 20, 1155)                # toy ring: generic NTT-Barrett path, 16K-point transforms
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE100 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        g.init_relin(ek_raw)
        T, REPS, lvl = 6, 4, 0
        logq, npr = o.logq(lvl), o.np_(lvl)
        ins = [[(_rand_crt(o, npr, 1000 + 10 * t + r), _rand_crt(o, npr, 2000 + 10 * t + r)) for r in range(REPS)] for t in range(T)]
        want = [[o.mul_relin_crt(a, b, lvl, ek) for a, b in ins[t]] for t in range(T)]
        # device buffers are prepared up front on torch's stream; the threads only call the C ABI
        bufs = []
        for t in range(T):
            per = []
            for a, b in ins[t]:
                per.append(dict(ca=gu.to_dev(a), cb=gu.to_dev(b), na=gu.empty_u64(npr, q.nttLen), nb=gu.empty_u64(npr, q.nttLen),
                                cr=gu.empty_u32(npr, q.crtLen), raw=gu.empty_u32(q.rawLen, o.words(lvl))))
            bufs.append(per)
        torch.cuda.synchronize()
        errors = []
        barrier = threading.Barrier(T)

        def work(t):
            try:
                st = ctypes.c_void_p()
                gu.ck(gu.lib.cuhe_hip_stream_create(0, ctypes.byref(st)))
                barrier.wait()
                for d in bufs[t]:
                    L = gu.lib
                    gu.ck(L.cuhe_hip_ntt(d["na"].data_ptr(), d["ca"].data_ptr(), logq, 0, st))
                    gu.ck(L.cuhe_hip_ntt(d["nb"].data_ptr(), d["cb"].data_ptr(), logq, 0, st))
                    gu.ck(L.cuhe_hip_ntt_mul(d["na"].data_ptr(), d["na"].data_ptr(), d["nb"].data_ptr(), logq, 0, st))
                    gu.ck(L.cuhe_hip_intt_mod(d["cr"].data_ptr(), d["na"].data_ptr(), logq, 0, st))
                    gu.ck(L.cuhe_hip_icrt(d["raw"].data_ptr(), d["cr"].data_ptr(), logq, 0, st))
                    gu.ck(L.cuhe_hip_relinearization(d["na"].data_ptr(), d["raw"].data_ptr(), lvl, 0, st))
                    gu.ck(L.cuhe_hip_intt_mod(d["cr"].data_ptr(), d["na"].data_ptr(), logq, 0, st))
                gu.ck(gu.lib.cuhe_hip_stream_sync(0, st))
                gu.ck(gu.lib.cuhe_hip_stream_destroy(0, st))
            except Exception as e:                      # surfaced in the main thread
                errors.append((t, repr(e)))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        for th in threads: th.start()
        for th in threads: th.join()
        assert not errors, errors
        for t in range(T):
            for r in range(REPS):
                assert np.array_equal(gu.host_u32(bufs[t][r]["cr"]), want[t][r]), (t, r)

        # second round: every thread issues ONE batched call over its REPS operands with the lanes forced on (groups of
        # four ciphertexts on the thread's own stream and its helper streams), several times, all threads at once
        B2 = 9
        gu.ck(gu.lib.cuhe_hip_set_relin_lanes(-3))
        stacks = []
        for t in range(T):
            na = gu.empty_u64(B2 * npr, g.ctlen); nb = gu.empty_u64(B2 * npr, g.ctlen)
            for i in range(B2):
                d = bufs[t][i % REPS]
                gu.ck(gu.lib.cuhe_hip_ct_ntt(na[i * npr:].data_ptr(), d["ca"].data_ptr(), logq, 0, None))
                gu.ck(gu.lib.cuhe_hip_ct_ntt(nb[i * npr:].data_ptr(), d["cb"].data_ptr(), logq, 0, None))
            stacks.append((na, nb, gu.empty_u32(B2 * npr, q.crtLen)))
        torch.cuda.synchronize()
        barrier2 = threading.Barrier(T)

        def work2(t):
            try:
                st = ctypes.c_void_p()
                gu.ck(gu.lib.cuhe_hip_stream_create(0, ctypes.byref(st)))
                barrier2.wait()
                na, nb, out = stacks[t]
                for _ in range(3):
                    gu.ck(gu.lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, B2, 0, st))
                gu.ck(gu.lib.cuhe_hip_stream_sync(0, st))
                gu.ck(gu.lib.cuhe_hip_stream_destroy(0, st))
            except Exception as e:
                errors.append((t, repr(e)))

        threads = [threading.Thread(target=work2, args=(t,)) for t in range(T)]
        for th in threads: th.start()
        for th in threads: th.join()
        gu.ck(gu.lib.cuhe_hip_set_relin_lanes(1))
        assert not errors, errors
        for t in range(T):
            got = gu.host_u32(stacks[t][2]).reshape(B2, npr, q.crtLen)
            for i in range(B2):
                assert np.array_equal(got[i], want[t][i % REPS]), ("batched, lanes", t, i)
    finally:
        g.close(); o.close()


def test_config4_relin_structured_keys_vs_python(gu):
    """BASELINE config 4 at full size (64K-point transforms, 48 CRT primes, 72 evaluation keys of 16-bit windows):
    relinearisation pinned by exact Python integers, independent of the oracle.  With the keys ek_j = s * 2^(w j) mod q0
    the key-switch sum  sum_j window_j(c) * ek_j  equals  c * s  modulo Phi and q (the windows recompose c), and for a
    sparse s the product c * s mod (x^n + 1) is a few negacyclic shifts of c -- cheap to compute exactly in Python.
    Covers the window extraction, the 72 window transforms, the 3456 key transforms of initRelinearization, the
    inner-product kernel and the fused INTT + reduction, at levels 0 and 5."""
    import oracle_lib as O
    g = gu.GpuCtx(25, 2, 16, 576, 24, 65536)
    try:
        q = g.prm
        assert (q.nttLen, q.numCrtPrime, q.numEvalKey, q.modLen) == (65536, 48, 72, 32768)
        n, w, K = q.modLen, q.logRelin, q.numEvalKey
        q0 = g.coeff_modulus(0)
        primes = g.crt_primes()
        rng = np.random.default_rng(4)
        terms = [(int(e), int.from_bytes(rng.bytes(150), "little") % q0) for e in (0, 1, 777, 20011, n - 1)]   # s = sum v x^e
        W0 = g.words(0)
        ek_raw = np.zeros((K, q.rawLen, W0), dtype=np.uint32)
        for j in range(K):
            for e, v in terms:
                ek_raw[j, e] = np.frombuffer(((v << (w * j)) % q0).to_bytes(4 * W0, "little"), dtype=np.uint32)
        g.init_relin(ek_raw)
        for lvl in (0, 5):
            ql, npr = g.coeff_modulus(lvl), g.np_(lvl)
            ct, cv = O.random_raw(q.rawLen, n, g.words(lvl), ql, 0xC400 + lvl)
            got = g.relin_crt(ct, lvl)                               # u32[npr][crtLen]
            # exact: (c * s mod x^n + 1) mod q_lvl, then its residues
            acc = [0] * n
            for e, v in terms:
                for i in range(n):
                    k = i + e
                    if k < n: acc[k] += cv[i] * v
                    else: acc[k - n] -= cv[i] * v
            want = np.zeros((npr, q.crtLen), dtype=np.uint32)
            for i in range(n):
                r = acc[i] % ql
                for t in range(npr):
                    want[t, i] = r % primes[t]
            assert np.array_equal(got, want), lvl
    finally:
        g.close()


@pytest.mark.parametrize("args", [(3, 2, 8, 40, 20, 1155), (3, 2, 16, 50, 25, 16384), (3, 2, 16, 25, 25, 21845),
                                  (5, 2, 1, 61, 20, 8191), (6, 2, 1, 61, 20, 8191), (3, 2, 16, 48, 24, 32768),
                                  (3, 2, 16, 50, 25, 32767), (3, 2, 16, 50, 25, 32749)],
                         ids=["toy1155-generic", "pow2_16384-fused", "prince_ring-generic", "dhs_simple-141keys-lds144K",
                              "w1-161keys-register-kernel", "pow2_32768-negacyclic", "phi32767-generic-64K", "prime32749-fold-64K"])
def test_mul_relin_batch_equals_single(gu, args):
    """cuhe_hip_mul_relin_batch (B independent cAnd + relin chains in one call: batch*np rows per stage, key values
    shared by four ciphertexts in the inner product) is bit-identical to B single-ciphertext sequences, which the other
    tests pin to the oracle; odd batch sizes exercise the tail of the ciphertext blocking, level 1 the prime tables.  The
    last two rings have 1-bit windows: 141 keys fill the LDS window tile to 144 KB (one workgroup per CU), 161 keys exceed
    it and take the register-blocked inner-product kernel."""
    import oracle_lib as O
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE200 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        g.init_relin(ek_raw)
        # groups of four ciphertexts go round-robin to `lanes` streams of the calling thread (forced for these small rings
        # by the negative count): 6 = two lanes, 13 = four groups on four lanes with a one-ciphertext tail, 9 = three
        # groups on two lanes (a lane used twice)
        for lvl, B, lanes in ((0, 1, -3), (0, 3, -3), (1, 6, -3), (1, 13, -4), (1, 9, -2)):
            gu.ck(gu.lib.cuhe_hip_set_relin_lanes(lanes))
            npr = o.np_(lvl)
            a = [_rand_crt(o, npr, 3000 + 10 * lvl + i) for i in range(B)]
            b = [_rand_crt(o, npr, 4000 + 10 * lvl + i) for i in range(B)]
            single = [g.mul_relin_crt(a[i], b[i], lvl) for i in range(B)]
            if B == 1:
                assert np.array_equal(single[0], o.mul_relin_crt(a[0], b[0], lvl, ek))      # the anchor itself, once
            na = gu.empty_u64(B * npr, g.ctlen); nb = gu.empty_u64(B * npr, g.ctlen)
            for i in range(B):
                gu.ck(gu.lib.cuhe_hip_ct_ntt(na[i * npr:].data_ptr(), gu.to_dev(a[i]).data_ptr(), o.logq(lvl), 0, None))
                gu.ck(gu.lib.cuhe_hip_ct_ntt(nb[i * npr:].data_ptr(), gu.to_dev(b[i]).data_ptr(), o.logq(lvl), 0, None))
            out = gu.empty_u32(B * npr, q.crtLen)
            gu.ck(gu.lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, B, 0, None))
            got = gu.host_u32(out).reshape(B, npr, q.crtLen)
            for i in range(B):
                assert np.array_equal(got[i], single[i]), (lvl, B, i)
            # cuhe_hip_relin_batch: relinearise the B results once more (CRT-domain input, in place) against the
            # single-ciphertext sequence icrt, relinearization, intt_mod
            again = [g.relin_crt(g.icrt(single[i], lvl), lvl) for i in range(B)]
            gu.ck(gu.lib.cuhe_hip_relin_batch(out.data_ptr(), out.data_ptr(), lvl, B, 0, None))
            got2 = gu.host_u32(out).reshape(B, npr, q.crtLen)
            for i in range(B):
                assert np.array_equal(got2[i], again[i]), ("relin_batch", lvl, B, i)
        gu.ck(gu.lib.cuhe_hip_set_relin_lanes(1))
        assert gu.lib.cuhe_hip_set_relin_lanes(0) != 0 and gu.lib.cuhe_hip_set_relin_lanes(5) != 0 and gu.lib.cuhe_hip_set_relin_lanes(-5) != 0
        assert gu.lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), 0, 0, 0, None) != 0      # batch < 1
        assert gu.lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), 99, 1, 0, None) != 0     # bad level
    finally:
        g.close(); o.close()


@pytest.mark.parametrize("args", [(3, 2, 8, 40, 20, 1155), (3, 2, 16, 48, 24, 32768), (25, 2, 16, 25, 25, 21845), (5, 2, 2, 28, 25, 16384),
                                  (12, 2, 4, 25, 25, 16384), (4, 2, 1, 61, 20, 8191), (5, 2, 1, 28, 25, 16384)],
                         ids=["toy1155-10keys", "pow2_32768-negacyclic-6keys", "prince-40keys-25primes", "64keys", "75keys-12primes",
                              "121keys-prime_m", "128keys"])
def test_matrix_core_inner_product_equals_valu_kernel(gu, args):
    """The key-switch inner product of the batched calls on the matrix cores (k_relin_mac_mfma: signed base-256 digits,
    int8 MFMA, exact int32 accumulation) against the VALU kernel (k_relin_mac_lds), which the other tests pin to the
    single-ciphertext sequence and the oracle: bit-identical CRT rows.  The rings cover every instantiated shape of the
    window dimension (tails of 8- and 16-byte lanes, one and two 64-window steps), prime tiles that are not full (4, 6,
    12, 25 primes), levels with fewer windows and primes than the key digits were laid out for, and batch sizes with an
    unfilled tile (8, 27 = 16 + 11) or a remainder that goes to the VALU kernel (21 = 16 + 5)."""
    import oracle_lib as O
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        rng = np.random.default_rng(0xE300)
        ek_raw = np.zeros((K, q.rawLen, W0), dtype=np.uint32)
        ek_raw[:, :q.modLen] = rng.integers(0, 1 << 32, (K, q.modLen, W0), dtype=np.uint64).astype(np.uint32)
        ek_raw[:, :, W0 - 1] &= (1 << max(0, (o.logq(0) - 1) % 32)) - 1        # below the coefficient modulus
        g.init_relin(ek_raw)
        for lvl, B in ((0, 16), (1, 8), (1, 21), (0, 27)):
            npr = o.np_(lvl)
            rows = np.concatenate([_rand_crt(o, npr, 5000 + 100 * lvl + i) for i in range(B)])
            src = gu.to_dev(rows)
            outs = []
            for mfma in (0, 8):                              # 8: the remainder of 21 = 16 + 5 goes to the VALU kernel, 27 = 16 + 11 is padded
                gu.ck(gu.lib.cuhe_hip_set_relin_mfma(mfma))
                out = gu.empty_u32(B * npr, q.crtLen)
                gu.ck(gu.lib.cuhe_hip_relin_batch(out.data_ptr(), src.data_ptr(), lvl, B, 0, None))
                outs.append(gu.host_u32(out).reshape(B, npr, q.crtLen))
            assert np.array_equal(outs[0], outs[1]), (lvl, B, [i for i in range(B) if not np.array_equal(outs[0][i], outs[1][i])][:8])
            one = g.relin_crt(g.icrt(rows[:npr], lvl), lvl)                     # the single-ciphertext sequence, once
            assert np.array_equal(outs[1][0], one), (lvl, B)
        gu.ck(gu.lib.cuhe_hip_set_relin_mfma(5))
        assert gu.lib.cuhe_hip_set_relin_mfma(-1) != 0
    finally:
        gu.lib.cuhe_hip_set_relin_mfma(5)
        g.close(); o.close()


@pytest.mark.parametrize("args", [(3, 2, 8, 40, 20, 1155), (3, 2, 16, 50, 25, 16384), (5, 2, 1, 61, 20, 8191), (3, 2, 16, 48, 24, 32768),
                                  (3, 2, 16, 50, 25, 32767), (3, 2, 16, 50, 25, 32749), (3, 2, 16, 50, 25, 16381)],
                         ids=["toy1155-generic", "pow2_16384-fused", "dhs_simple-prime_m", "pow2_32768-negacyclic",
                              "phi32767-generic-64K", "prime32749-fold-64K", "prime16381-fold-32K"])
def test_mul_raw_batch_equals_single(gu, args):
    """cuhe_hip_mul_raw_batch (B full multiplications raw -> raw per call) against the oracle's mulZZX-at-the-raw-level
    for every ciphertext of the batch, on the three reduction kinds, at two levels, with odd batch sizes."""
    import oracle_lib as O
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        for lvl, B in ((0, 1), (0, 3), (1, 4)):
            W, M = o.words(lvl), o.coeff_modulus(lvl)
            a = [O.random_raw(q.rawLen, q.modLen, W, M, 5000 + 10 * lvl + i)[0] for i in range(B)]
            b = [O.random_raw(q.rawLen, q.modLen, W, M, 6000 + 10 * lvl + i)[0] for i in range(B)]
            da, db = gu.to_dev(np.stack(a).reshape(B * q.rawLen, W)), gu.to_dev(np.stack(b).reshape(B * q.rawLen, W))
            out = gu.empty_u32(B * q.rawLen, W)
            gu.ck(gu.lib.cuhe_hip_mul_raw_batch(out.data_ptr(), da.data_ptr(), db.data_ptr(), lvl, B, 0, None))
            got = gu.host_u32(out).reshape(B, q.rawLen, W)
            for i in range(B):
                assert np.array_equal(got[i], o.mul_raw(a[i], b[i], lvl)), (lvl, B, i)
        assert gu.lib.cuhe_hip_mul_raw_batch(out.data_ptr(), da.data_ptr(), db.data_ptr(), 0, 0, 0, None) != 0
    finally:
        g.close(); o.close()


@pytest.mark.parametrize("args", [(3, 2, 8, 40, 20, 1155), (3, 2, 16, 50, 25, 16384), (3, 2, 16, 48, 24, 32768),
                                  (3, 2, 16, 50, 25, 32767), (3, 2, 16, 50, 25, 32749)],
                         ids=["toy1155-generic", "pow2_16384-fused", "pow2_32768-negacyclic", "phi32767-generic-64K", "prime32749-fold-64K"])
def test_array_gates_equal_single_gates(gu, args):
    """gates on arrays of ciphertexts (cuhe_hip_ntt_mul_pairs, intt_mod_batch, crt_mod_switch_batch, crt_combine) against the
    per-ciphertext entry points they stand for (ntt_mul, intt_mod, crt_mod_switch, crt_add / crt_add_int), which the
    other tests pin to the oracle."""
    import torch
    import oracle_lib as O
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        lvl, B = 0, 5
        npr, logq = o.np_(lvl), o.logq(lvl)
        cts = [_rand_crt(o, npr, 7000 + i) for i in range(B)]
        arr = gu.to_dev(np.concatenate(cts))                              # u32[B*np][crtLen]
        # forward transforms of the whole array, then products over index pairs
        ntt = gu.empty_u64(B * npr, g.ctlen)
        gu.ck(gu.lib.cuhe_hip_ntt_rows(ntt.data_ptr(), arr.data_ptr(), B * npr, 0, None))
        pairs = [(0, 1), (2, 2), (4, 0), (3, 1)]
        ia = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=gu.DEV)
        ib = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=gu.DEV)
        prod = gu.empty_u64(len(pairs) * npr, g.ctlen)
        gu.ck(gu.lib.cuhe_hip_ntt_mul_pairs(prod.data_ptr(), ntt.data_ptr(), ia.data_ptr(), ib.data_ptr(), len(pairs), npr, 0, None))
        red = gu.empty_u32(len(pairs) * npr, q.crtLen)
        gu.ck(gu.lib.cuhe_hip_intt_mod_batch(red.data_ptr(), prod.data_ptr(), lvl, len(pairs), 0, None))
        got = gu.host_u32(red).reshape(len(pairs), npr, q.crtLen)
        for t, (x, y) in enumerate(pairs):
            want = g.intt_mod(o.ntt_mul(o.ntt(cts[x]), o.ntt(cts[y])), lvl)
            assert np.array_equal(got[t], want), ("pairs", t)
        # modulus switching of the whole array: packed [B][np-1][crtLen]
        ms = gu.empty_u32(B * (npr - 1), q.crtLen)
        gu.ck(gu.lib.cuhe_hip_crt_mod_switch_batch(ms.data_ptr(), arr.data_ptr(), lvl, B, 0, None))
        gotms = gu.host_u32(ms).reshape(B, npr - 1, q.crtLen)
        for i in range(B):
            assert np.array_equal(gotms[i], o.modswitch(cts[i])[:npr - 1]), ("modswitch", i)
        # index-list sums with constants: out0 = c0 + c3 + 1, out1 = c1, out2 = c2 + c4 + c0 (second source array = products)
        lists = [[0, 3], [1], [2, 4, 0, B + 1]]
        consts = [1, 0, 0]
        off = torch.tensor(np.cumsum([0] + [len(l) for l in lists]), dtype=torch.int32, device=gu.DEV)
        lst = torch.tensor([e for l in lists for e in l], dtype=torch.int32, device=gu.DEV)
        cst = torch.tensor(consts, dtype=torch.int32, device=gu.DEV)
        out = gu.empty_u32(len(lists) * npr, q.crtLen)
        gu.ck(gu.lib.cuhe_hip_crt_combine(out.data_ptr(), arr.data_ptr(), B, red.data_ptr(), off.data_ptr(), lst.data_ptr(), cst.data_ptr(),
                                        len(lists), lvl, 0, None))
        gotc = gu.host_u32(out).reshape(len(lists), npr, q.crtLen)
        primes = np.array(g.crt_primes()[:npr], dtype=np.uint64).reshape(npr, 1)
        pool = cts + [got[t] for t in range(len(pairs))]
        for oi, (l, c) in enumerate(zip(lists, consts)):
            acc = np.zeros((npr, q.crtLen), dtype=np.uint64)
            for e in l:
                acc += pool[e].astype(np.uint64)
            acc[:, 0] += c
            assert np.array_equal(gotc[oi], (acc % primes).astype(np.uint32)), ("combine", oi)
    finally:
        g.close(); o.close()


def test_config2_roundtrip_8_primes(gu):
    """BASELINE config 2: N = 2^14 (32K-point transforms), 8 CRT primes: forward + inverse round trip is the identity and
    the forward transform equals the oracle's, for residues spanning the whole range [0, p_i)."""
    import oracle_lib as O
    args = (1, 2, 16, 200, 25, 32768)
    g, o = gu.GpuCtx(*args), O.Ctx(*args)
    try:
        q = o.prm
        assert (q.modLen, q.nttLen, q.numCrtPrime) == (16384, 32768, 8)
        a = _rand_crt(o, 8, 77, True)
        X = g.ntt(a, 0)
        assert np.array_equal(X, o.ntt(a))
        assert np.array_equal(g.intt(X, 0), a)
    finally:
        g.close(); o.close()


def test_negacyclic_equals_cyclic_representation(gu):
    """x^16384 + 1 with 24-bit primes: the negacyclic ciphertext domain (transforms of modLen points, no reduction step,
    keys of half the size) against the reference's cyclic representation forced on the same ring and against the oracle's
    reference-shaped path: full multiply, multiply + relinearise, relinearisation alone, a round trip and an addition in
    the ct domain, at the first and the last level."""
    import oracle_lib as O
    args = PSETS["pow2_32768"]
    o = O.Ctx(*args)
    res = {}
    try:
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE700 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        for nc in (True, False):
            g = gu.GpuCtx(*args, negacyclic=nc)
            try:
                assert g.nc == nc and g.ctlen == (q.modLen if nc else q.nttLen)
                g.init_relin(ek_raw)
                for lvl in (0, q.depth - 1):
                    npr, W, M = o.np_(lvl), o.words(lvl), o.coeff_modulus(lvl)
                    a, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xA700 + lvl)
                    b, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xB700 + lvl)
                    ca, cb = _rand_crt(o, npr, 71 + lvl), _rand_crt(o, npr, 72 + lvl, full=True)      # full: every residue p - 1
                    out = dict(mul=g.mul_raw(a, b, lvl), mulfull=g.mul_relin_crt(cb, cb, lvl), mr=g.mul_relin_crt(ca, cb, lvl),
                               relin=g.relin_crt(a, lvl), rt=g.ct_intt(g.ct_ntt(ca, lvl), lvl, False),
                               add=g.ct_intt(g.ct_add(g.ct_ntt(ca, lvl), g.ct_ntt(cb, lvl), lvl), lvl, False))
                    if nc:
                        assert np.array_equal(out["mul"], o.mul_raw(a, b, lvl))
                        assert np.array_equal(out["mr"], o.mul_relin_crt(ca, cb, lvl, ek))
                        assert np.array_equal(out["mulfull"], o.mul_relin_crt(cb, cb, lvl, ek))
                        assert np.array_equal(out["relin"], o.intt_mod(o.relin(a, lvl, ek)))
                        assert np.array_equal(out["rt"], ca) and np.array_equal(out["add"], o.crt_add(ca, cb))
                        assert np.array_equal(g.mul_raw(a, b, lvl, cyclic_api=True), out["mul"])         # the reference-contract entry points still work
                        res[lvl] = out
                    else:
                        for k, v in out.items():
                            assert np.array_equal(v, res[lvl][k]), (k, lvl)
            finally:
                g.close()
    finally:
        o.close()


def test_degree_65536_ring(gu):
    """x^65536 + 1 (m = 131072, BASELINE config 4 read literally: N = 2^16, 48 CRT primes): beyond the reference's
    65536-point limit, reachable only through 64K-point NEGACYCLIC transforms (primes of 23 bits: 2 n p^2 < P).
    Dense random products against the oracle's negacyclic restatement on every prime; the full multiply and the
    relinearisation against exact Python integers (sparse second operand / structured keys, as in
    test_config4_relin_structured_keys_vs_python); the batched call against the single sequence; the cyclic entry points
    must refuse."""
    import oracle_lib as O
    g = gu.GpuCtx(25, 2, 16, 552, 23, 131072)
    try:
        q = g.prm
        assert g.nc and g.ctlen == 65536
        assert (q.modLen, q.crtLen, q.nttLen, q.numCrtPrime, q.numEvalKey, q.logCrtPrime) == (65536, 65536, 65536, 48, 69, 23)
        n, w, K = q.modLen, q.logRelin, q.numEvalKey
        primes = g.crt_primes()
        assert 2 * n * (max(primes) - 1) ** 2 < O.P
        lib, ck = gu.lib, gu.ck
        # ---- dense products per prime vs the oracle (level 0: 48 primes)
        lvl = 0
        npr, W, M = g.np_(lvl), g.words(lvl), g.coeff_modulus(lvl)
        a, av = O.random_raw(q.rawLen, n, W, M, 0x6501)
        b, bv = O.random_raw(q.rawLen, n, W, M, 0x6502)
        ca, cb = g.crt(a, lvl), g.crt(b, lvl)
        for i in (0, 7, 47):                                             # the CRT rows themselves, on a stride, vs Python
            assert [int(v) for v in ca[i, ::4099]] == [x % primes[i] for x in av[::4099]]
        prod = g.ct_intt(g.ct_mul(g.ct_ntt(ca, lvl), g.ct_ntt(cb, lvl), lvl), lvl, True)
        for i in range(npr):
            assert np.array_equal(prod[i], O.negacyclic_mul_modp(ca[i], cb[i], primes[i])), i
        full = np.stack([np.full(n, p - 1, dtype=np.uint32) for p in primes[:npr]])          # largest magnitudes
        pf = g.ct_intt(g.ct_mul(g.ct_ntt(full, lvl), g.ct_ntt(full, lvl), lvl), lvl, True)
        for i in (0, 23, 47):
            assert np.array_equal(pf[i], O.negacyclic_mul_modp(full[i], full[i], primes[i])), i
        # ---- full multiply raw -> raw vs exact integers: b sparse
        terms = [(0, 3), (1, M - 1), (40000, 0x1234567890ABCDEF % M), (n - 1, M // 3)]
        bs = [0] * n
        for e, v in terms: bs[e] = v
        got = O.raw_to_ints(g.mul_raw(a, O.ints_to_raw(bs, q.rawLen, W), lvl), n)
        acc = [0] * n
        for e, v in terms:
            for i in range(n):
                k = i + e
                if k < n: acc[k] += av[i] * v
                else: acc[k - n] -= av[i] * v
        assert got == [x % M for x in acc]
        # ---- relinearisation with structured keys ek_j = s * 2^(w j) mod q0, s sparse: the key-switch sum is c * s
        q0 = g.coeff_modulus(0)
        rng = np.random.default_rng(65)
        sterms = [(int(e), int.from_bytes(rng.bytes(150), "little") % q0) for e in (0, 5, 33333, n - 1)]
        W0 = g.words(0)
        ek_raw = np.zeros((K, q.rawLen, W0), dtype=np.uint32)
        for j in range(K):
            for e, v in sterms:
                ek_raw[j, e] = np.frombuffer(((v << (w * j)) % q0).to_bytes(4 * W0, "little"), dtype=np.uint32)
        g.init_relin(ek_raw)
        for lvl in (0, 7):
            ql, npr = g.coeff_modulus(lvl), g.np_(lvl)
            ct, cv = O.random_raw(q.rawLen, n, g.words(lvl), ql, 0x6510 + lvl)
            got = g.relin_crt(ct, lvl)
            acc = [0] * n
            for e, v in sterms:
                for i in range(n):
                    k = i + e
                    if k < n: acc[k] += cv[i] * v
                    else: acc[k - n] -= cv[i] * v
            for t in (0, npr // 2, npr - 1):
                assert [int(x) for x in got[t]] == [(r % ql) % primes[t] for r in acc], (lvl, t)
        # ---- batched multiply + relinearise equals the single sequence
        lvl, B = 1, 3
        npr = g.np_(lvl)
        xs = [g.crt(O.random_raw(q.rawLen, n, g.words(lvl), g.coeff_modulus(lvl), 0x6520 + i)[0], lvl) for i in range(2 * B)]
        single = [g.mul_relin_crt(xs[2 * i], xs[2 * i + 1], lvl) for i in range(B)]
        na, nb = gu.empty_u64(B * npr, n), gu.empty_u64(B * npr, n)
        for i in range(B):
            ck(lib.cuhe_hip_ct_ntt(na[i * npr:].data_ptr(), gu.to_dev(xs[2 * i]).data_ptr(), g.logq(lvl), 0, None))
            ck(lib.cuhe_hip_ct_ntt(nb[i * npr:].data_ptr(), gu.to_dev(xs[2 * i + 1]).data_ptr(), g.logq(lvl), 0, None))
        out = gu.empty_u32(B * npr, q.crtLen)
        ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, B, 0, None))
        got = gu.host_u32(out).reshape(B, npr, q.crtLen)
        for i in range(B):
            assert np.array_equal(got[i], single[i]), i
        # ---- the reference's cyclic transforms do not exist on this ring
        assert lib.cuhe_hip_ntt(na.data_ptr(), gu.to_dev(xs[0]).data_ptr(), g.logq(lvl), 0, None) != 0
        assert b"negacyclic" in lib.cuhe_hip_last_error()
        assert lib.cuhe_hip_intt_mod(out.data_ptr(), na.data_ptr(), g.logq(lvl), 0, None) != 0
    finally:
        g.close()


def test_sharded_multiply_through_the_c_abi(gu):
    """CRT-prime-sharded cAnd + relin behind the C ABI (include/cuhe_hip.h, "multi-GPU"): (1) the one-process-per-GPU
    form on a communicator of one rank -- RCCL is opened, the unique id made and the communicator initialised, the
    all-gather is a no-op: the chain must equal the unsharded one; (2) the in-process form over three VIRTUAL devices
    (own context each on the one GPU): rows travel by peer copies ordered by events; equal to the single-device result,
    called repeatedly and from two different home devices.  Both on a cyclic and on a negacyclic ring."""
    import ctypes
    import oracle_lib as O
    lib, ck = gu.lib, gu.ck
    for args in (PSETS["toy1155"], PSETS["pow2_32768"]):
        o = O.Ctx(*args)
        q = o.prm
        K, W0, M0 = q.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xE900 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        try:
            # (1) communicator of one rank
            g = gu.GpuCtx(*args)
            try:
                g.init_relin(ek_raw)
                uid = (ctypes.c_uint8 * 128)()
                ck(lib.cuhe_hip_comm_unique_id(uid))
                ck(lib.cuhe_hip_comm_init(1, 0, uid))
                assert lib.cuhe_hip_comm_size() == 1 and lib.cuhe_hip_comm_rank() == 0
                f, c = ctypes.c_int(), ctypes.c_int()
                ck(lib.cuhe_hip_shard_bounds(1, 3, 1, ctypes.byref(f), ctypes.byref(c)))
                from cuhe_amd.sharded import shard_bounds
                assert (f.value, c.value) == shard_bounds(o.np_(1), 3, 1)
                for lvl in (0, 1):
                    npr = o.np_(lvl)
                    a, b = _rand_crt(o, npr, 91 + lvl), _rand_crt(o, npr, 92 + lvl)
                    na, nb = gu.to_dev(g.ct_ntt(a, lvl)), gu.to_dev(g.ct_ntt(b, lvl))
                    out = gu.empty_u32(npr, q.crtLen)
                    ck(lib.cuhe_hip_mul_relin_sharded(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, 0, None))
                    assert np.array_equal(gu.host_u32(out), o.mul_relin_crt(a, b, lvl, ek)), lvl
                ck(lib.cuhe_hip_comm_destroy())
            finally:
                lib.cuhe_hip_comm_destroy(); g.close()
            # (2) three virtual devices in this process
            lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
            ck(lib.cuhe_hip_set_virtual_devices(1))
            try:
                ck(lib.cuhe_hip_set_parameters(*args))
                ck(lib.cuhe_hip_multi_gpus(3))
                ck(lib.cuhe_hip_init(None, 0))
                ekc = np.ascontiguousarray(ek_raw, dtype=np.uint32)
                ck(lib.cuhe_hip_init_relin(ekc.ctypes.data_as(ctypes.c_void_p)))
                ctlen = lib.cuhe_hip_ct_len()
                for lvl in (0, 1):
                    npr, logq = o.np_(lvl), o.logq(lvl)
                    a, b = _rand_crt(o, npr, 93 + lvl), _rand_crt(o, npr, 94 + lvl)
                    want = o.mul_relin_crt(a, b, lvl, ek)
                    for dev0 in (0, 2):
                        na, nb = gu.empty_u64(npr, ctlen), gu.empty_u64(npr, ctlen)
                        ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), gu.to_dev(a).data_ptr(), logq, dev0, None))
                        ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), gu.to_dev(b).data_ptr(), logq, dev0, None))
                        out = gu.empty_u32(npr, q.crtLen)
                        for rep in range(3):
                            out.zero_()
                            ck(lib.cuhe_hip_mul_relin_sharded_inproc(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, dev0, None))
                            ck(lib.cuhe_hip_stream_sync(dev0, None))
                            assert np.array_equal(gu.host_u32(out), want), (lvl, dev0, rep)
            finally:
                lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters(); lib.cuhe_hip_set_virtual_devices(0); lib.cuhe_hip_multi_gpus(1)
        finally:
            o.close()


@pytest.mark.parametrize("name,args", [("toy1155", (3, 2, 8, 40, 20, 1155)), ("prince_small", (3, 2, 16, 25, 25, 21845)),
                                       ("pow2_16384", (3, 2, 16, 50, 25, 16384)), ("pow2_32768", (3, 2, 16, 48, 24, 32768)),
                                       ("c3_65536", (9, 2, 16, 576, 24, 65536)), ("n65536", (3, 2, 16, 46, 23, 131072)),
                                       ("phi32767", (3, 2, 16, 50, 25, 32767)), ("prime32749", (3, 2, 16, 50, 25, 32749)),
                                       ("prime16381", (3, 2, 16, 50, 25, 16381))])
def test_low_latency_kernels_equal_throughput_kernels(gu, name, args):
    """Every form of the transform kernels (two passes with 16 values per thread: throughput; 4 values per thread, radix-4
    through LDS: low latency, taken below cuhe_hip_set_ll_rows rows; ONE workgroup per 8K / 16K / 32K-point sub-transform,
    cuhe_hip_set_onewg, forced here whatever the row count; its persistent form on a large batch of 64K-point rows)
    gives the same bits on every source / store variant: zero-padded
    and full forward transforms (all three lengths), window transforms, index-negated inverses with `% p` and a ragged store
    count, the fused x^n+1 reduction, the folded generic reduction's two epilogues, table products, the negacyclic twist
    and untwist, the product-on-load inverse of the batched chain.  One run of each operation per form, compared with each other and (once) with the oracle."""
    import oracle_lib as O
    lib, ck = gu.lib, gu.ck
    ALL, NONE = 1 << 30, 0
    o = O.Ctx(*args) if name != "n65536" else None
    res = {}
    try:
        # (rows64k = 2 with the one-workgroup form forced: the persistent kernels also take the negacyclic 64K-point rows of x^65536 + 1)
        # split: negacyclic rows as two half-length sub-transforms (cuhe_hip_set_onewg_split; 1 = the default, inverse rows of 32K points;
        # 2 with the one-workgroup form forced: every split kernel -- forward rows of 32K, inverse rows of 32K and 64K points)
        forms = [(NONE, 0, 0, 0), (ALL, 0, 0, 0), (NONE, 2, 1, 0), (NONE, 2, 1, 1)]
        if name in ("n65536", "c3_65536"): forms.append((NONE, 2, 2, 1))
        if name in ("n65536", "c3_65536"): forms.append((NONE, 2, 1, 2))           # (rings whose ciphertext domain has 32K / 64K points)
        for form, onewg, r64, split in forms:
            g = gu.GpuCtx(*args)
            ck(lib.cuhe_hip_set_ll_rows(form))
            ck(lib.cuhe_hip_set_onewg(onewg, r64))
            ck(lib.cuhe_hip_set_onewg_split(split))
            try:
                q = g.prm
                K, W0, M0 = q.numEvalKey, g.words(0), g.coeff_modulus(0)
                ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, 0xEA00 + j)[0] for j in range(K)])
                g.init_relin(ek_raw)
                out = {}
                for lvl in (0, q.depth - 1):
                    npr, W, M = g.np_(lvl), g.words(lvl), g.coeff_modulus(lvl)
                    a, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xA900 + lvl)
                    b, _ = O.random_raw(q.rawLen, q.modLen, W, M, 0xB900 + lvl)
                    ca, cb = g.crt(a, lvl), g.crt(b, lvl)
                    out[("mul", lvl)] = g.mul_raw(a, b, lvl)
                    out[("mulrelin", lvl)] = g.mul_relin_crt(ca, cb, lvl)
                    out[("ctrt", lvl)] = g.ct_intt(g.ct_ntt(ca, lvl), lvl, False)
                    # the batched chain: inverse transforms that form the product of their two operands on load
                    nab = gu.empty_u64(2 * npr, g.ctlen); nbb = gu.empty_u64(2 * npr, g.ctlen)
                    for i, (x, y) in enumerate(((ca, cb), (cb, cb))):
                        ck(lib.cuhe_hip_ct_ntt(nab[i * npr:].data_ptr(), gu.to_dev(x).data_ptr(), g.logq(lvl), 0, None))
                        ck(lib.cuhe_hip_ct_ntt(nbb[i * npr:].data_ptr(), gu.to_dev(y).data_ptr(), g.logq(lvl), 0, None))
                    ob = gu.empty_u32(2 * npr, q.crtLen)
                    ck(lib.cuhe_hip_mul_relin_batch(ob.data_ptr(), nab.data_ptr(), nbb.data_ptr(), lvl, 2, 0, None))
                    out[("mulrelin_batch", lvl)] = gu.host_u32(ob)
                    assert np.array_equal(out[("mulrelin_batch", lvl)][:npr], out[("mulrelin", lvl)]), (name, lvl, form)
                    if name != "n65536":
                        X = g.ntt(ca, lvl)
                        out[("ntt", lvl)] = X
                        out[("intt", lvl)] = g.intt(X, lvl)
                        out[("inttdd", lvl)] = g.intt_double_deg(g.ntt_mul(X, g.ntt(cb, lvl), lvl), lvl)
                        out[("inttmod", lvl)] = g.intt_mod(g.ntt_mul(X, g.ntt(cb, lvl), lvl), lvl)
                        out[("nttw", lvl)] = g.nttw(a, lvl)
                        gu.ck(lib.cuhe_hip_force_generic_reduce(1))
                        out[("generic", lvl)] = g.intt_mod(g.ntt_mul(X, g.ntt(cb, lvl), lvl), lvl)
                        gu.ck(lib.cuhe_hip_force_generic_reduce(0))
                        d = gu.empty_u32(3, q.nttLen)                       # ragged store count, prime offset
                        nst = q.nttLen // 3 + 5
                        ck(lib.cuhe_hip_ntt_inv_batched(d.data_ptr(), gu.to_dev(X[:3] if npr >= 3 else np.repeat(X[:1], 3, 0)).data_ptr(),
                                                        q.nttLen, 3, q.nttLen, nst, 0 if npr < 4 else 1, 0, None))
                        out[("ragged", lvl)] = gu.host_u32(d)
                if form == NONE and onewg == 0:
                    res = out
                    if o is not None:
                        a0 = O.random_raw(q.rawLen, q.modLen, W0, M0, 0xA900)[0]
                        b0 = O.random_raw(q.rawLen, q.modLen, W0, M0, 0xB900)[0]
                        assert np.array_equal(out[("mul", 0)], o.mul_raw(a0, b0, 0))
                else:
                    for k, v in out.items():
                        assert np.array_equal(v, res[k]), (name, k)
            finally:
                ck(lib.cuhe_hip_set_ll_rows(24))
                ck(lib.cuhe_hip_set_onewg(1, 2))
                ck(lib.cuhe_hip_set_onewg_split(1))
                g.close()
        # the standalone batched forward entry point at all three lengths, odd batch
        for length in (16384, 32768, 65536):
            x = np.stack([O.splitmix_u32_below(length // 2, 0xFFFFFFFF, 900 + b) for b in range(5)])
            ck(lib.cuhe_hip_ntt_prepare(length, 0))
            got = []
            for form, onewg in ((NONE, 0), (ALL, 0), (NONE, 2)):
                ck(lib.cuhe_hip_set_ll_rows(form))
                ck(lib.cuhe_hip_set_onewg(onewg, 1))
                dX = gu.empty_u64(5, length)
                ck(lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), gu.to_dev(x).data_ptr(), length, 5, length // 2, 0, None))
                got.append(gu.host_u64(dX))
            ck(lib.cuhe_hip_set_ll_rows(24))
            ck(lib.cuhe_hip_set_onewg(1, 2))
            assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2]), length
            assert np.array_equal(got[1][4], O.ntt_ext(x[4], length)), length
        if name == "toy1155":
            # the persistent form of the 64K-point rows needs a batch that gives every workgroup two halves; 301 rows: a
            # last group of 8 with padding items.  Against the two-pass kernels on every row, against the oracle on three.
            length, batch = 65536, 301
            x = np.stack([O.splitmix_u32_below(length // 2, 0xFFFFFFFF, 7000 + b) for b in range(batch)])
            dx = gu.to_dev(x)
            got = []
            import ctypes as C
            forms = []
            for onewg, r64 in ((0, 0), (2, 2), (1, 1)):               # (2, 2): the persistent form whatever the row count (the default takes it from 2560 rows on)
                ck(lib.cuhe_hip_set_onewg(onewg, r64))
                dX = gu.empty_u64(batch, length)
                for _ in range(2):                                      # twice: the second call re-uses the tables
                    ck(lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), dx.data_ptr(), length, batch, length // 2, 0, None))
                got.append(gu.host_u64(dX))
                info = C.create_string_buffer(256)
                ck(lib.cuhe_hip_last_dispatch_info(0, info, 256))
                forms.append(info.value.decode())
            assert forms[0].startswith("two-pass pair") and forms[1].startswith("persistent") and forms[2].startswith("one workgroup per half"), forms
            assert "301 rows of 65536 points" in forms[1] and "given up by 0 workgroups" in forms[1], forms[1]
            ck(lib.cuhe_hip_set_onewg(1, 0))
            assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
            for b in (0, 150, 300):
                assert np.array_equal(got[1][b], O.ntt_ext(x[b], length)), b
            # rows that are not 16-byte aligned cannot be fetched by LDS-DMA: the default policy takes the two-pass kernels for them
            ck(lib.cuhe_hip_set_onewg(1, 2))
            flat = gu.to_dev(np.concatenate((np.zeros(1, dtype=np.uint32), x.reshape(-1))))
            dX = gu.empty_u64(batch, length)
            ck(lib.cuhe_hip_ntt_fwd_batched(dX.data_ptr(), flat.data_ptr() + 4, length, batch, length // 2, 0, None))
            assert np.array_equal(gu.host_u64(dX), got[0])
            ck(lib.cuhe_hip_set_onewg(1, 2))
    finally:
        ck(lib.cuhe_hip_set_ll_rows(24))
        ck(lib.cuhe_hip_set_onewg(1, 2))
        if o is not None:
            o.close()


@pytest.mark.parametrize("name", ["toy1155", "dhs_simple", "prince_small", "pow2_32768", "phi32767", "prime32749"])
def test_transforms_of_separately_owned_blocks(gu, name):
    """cuhe_hip_ct_ntt_list / cuhe_hip_ct_intt_list (round 5: what the gate scheduler's batches call instead of gather + array transform +
    scatter): the one-workgroup kernels address the rows of every ciphertext inside that ciphertext's own block.  140 ciphertexts (up to
    128 blocks per launch) whose blocks lie in a shuffled order inside one pool (offsets of both signs) against the per-ciphertext transforms
    (cuhe_hip_ct_ntt) and the array forms on gathered rows (cuhe_hip_intt_batch / cuhe_hip_intt_mod_batch), bit for bit; products too; a call
    of 3 ciphertexts (takes the two-pass pair on the long rings: the entry points gather through their own scratch) and the forced
    gather form (cuhe_hip_set_row_lists(0)) give the same rows."""
    import ctypes as C
    lib, ck = gu.lib, gu.ck
    g = gu.GpuCtx(*PSETS[name])
    try:
        q = g.prm
        rng = np.random.default_rng(91)
        plist = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        for lvl, n in ((0, 70), (1, 3)):
            npr, logq, ctlen = g.np_(lvl), g.logq(lvl), g.ctlen
            crt = [np.zeros((npr, q.crtLen), dtype=np.uint32) for _ in range(2 * n)]
            for a in crt:
                for t in range(npr):
                    a[t, :q.modLen] = rng.integers(0, g.primes[t], q.modLen, dtype=np.uint32)
            # blocks in shuffled slots of one pool, a gap of one block between neighbours
            cpool, npool = gu.empty_u32(4 * n, npr, q.crtLen), gu.empty_u64(4 * n, npr, ctlen)
            slots = rng.permutation(2 * n)
            dcrt = [cpool[2 * int(sl)] for sl in slots]
            dntt = [npool[2 * int(sl)] for sl in slots]
            for t, a in zip(dcrt, crt):
                t.copy_(gu.to_dev(a))
            direct = C.c_int(-1)
            ck(lib.cuhe_hip_ct_ntt_list(plist(dntt), plist(dcrt), 2 * n, lvl, 0, None, C.byref(direct)))
            assert direct.value in (0, min(2 * n, 128), 2 * n), direct.value
            if name == "prince_small" and n == 70:
                assert direct.value == 128, direct.value              # 128 blocks x 3 rows of 32K points fill the chip (the form the scheduler's batches rely
                                                                      # on); the 12 ciphertexts left over are too few: gathered
            if name == "prince_small" and n == 3:
                assert direct.value == 0                              # the two-pass pair: through the entry point's own scratch
            want = gu.empty_u64(npr, ctlen)
            for i in sorted(set([0, 1, n // 2, n - 1, n, 2 * n - 1])):
                ck(lib.cuhe_hip_ct_ntt(want.data_ptr(), dcrt[i].data_ptr(), logq, 0, None))
                assert np.array_equal(gu.host_u64(dntt[i]), gu.host_u64(want)), (lvl, i)
            # ---- n2c of non-products: the round trip, and the array form on gathered rows
            back = gu.empty_u32(2 * n, npr, q.crtLen); back.zero_()
            ck(lib.cuhe_hip_ct_intt_list(back.data_ptr(), plist(dntt), 2 * n, lvl, 0, 0, None, C.byref(direct)))
            got = gu.host_u32(back)
            for i in range(2 * n):
                assert np.array_equal(got[i][:, :q.modLen], crt[i][:, :q.modLen]), (lvl, i)
            narr = gu.empty_u64(2 * n, npr, ctlen)
            ck(lib.cuhe_hip_gather_blocks(narr.data_ptr(), plist(dntt), 2 * n, npr * ctlen * 8, 0, None))
            ref = gu.empty_u32(2 * n, npr, q.crtLen); ref.zero_()
            ck(lib.cuhe_hip_intt_batch(ref.data_ptr(), narr.data_ptr(), lvl, 2 * n, 0, None))
            assert np.array_equal(got, gu.host_u32(ref))
            # ---- products: z_i = a_i * b_i in their own blocks, n2c with the reduction modulo the polynomial modulus
            zpool = gu.empty_u64(2 * n, npr, ctlen)
            z = [zpool[2 * int(k)] for k in rng.permutation(n)]
            ck(lib.cuhe_hip_ct_binop_list(1, plist(z), plist(dntt[:n]), plist(dntt[n:]), n, logq, 0, None))
            pr = gu.empty_u32(n, npr, q.crtLen); pr.zero_()
            ck(lib.cuhe_hip_ct_intt_list(pr.data_ptr(), plist(z), n, lvl, 1, 0, None, C.byref(direct)))
            zarr = gu.empty_u64(n, npr, ctlen)
            ck(lib.cuhe_hip_gather_blocks(zarr.data_ptr(), plist(z), n, npr * ctlen * 8, 0, None))
            pref = gu.empty_u32(n, npr, q.crtLen); pref.zero_()
            ck(lib.cuhe_hip_intt_mod_batch(pref.data_ptr(), zarr.data_ptr(), lvl, n, 0, None))
            assert np.array_equal(gu.host_u32(pr), gu.host_u32(pref)), lvl
            # ---- the forced gather form
            ck(lib.cuhe_hip_set_row_lists(0))
            try:
                pr2 = gu.empty_u32(n, npr, q.crtLen); pr2.zero_()
                ck(lib.cuhe_hip_ct_intt_list(pr2.data_ptr(), plist(z), n, lvl, 1, 0, None, C.byref(direct)))
                assert direct.value == 0 and np.array_equal(gu.host_u32(pr2), gu.host_u32(pref))
                again = [gu.empty_u64(npr, ctlen) for _ in range(2 * n)]
                ck(lib.cuhe_hip_ct_ntt_list(plist(again), plist(dcrt), 2 * n, lvl, 0, None, C.byref(direct)))
                assert direct.value == 0
                for i in (0, n, 2 * n - 1):
                    assert np.array_equal(gu.host_u64(again[i]), gu.host_u64(dntt[i])), (lvl, i)
            finally:
                ck(lib.cuhe_hip_set_row_lists(1))
            bad = (C.c_void_p * 2)(dntt[0].data_ptr(), None)
            assert lib.cuhe_hip_ct_intt_list(back.data_ptr(), bad, 2, lvl, 0, 0, None, None) != 0
        assert lib.cuhe_hip_ct_ntt_list(plist(dntt), plist(dcrt), 2, q.depth, 0, None, None) != 0          # no such level
    finally:
        g.close()


@pytest.mark.parametrize("name", ["toy1155", "pow2_32768", "phi32767"])
def test_list_block_and_event_entry_points(gu, name):
    """The round-4 additions to the C ABI that the C++ layer's gate scheduler is built on, each against the entry point it
    generalises: gather / scatter / copy of separately owned blocks through pointer lists, cAnd / cXor over lists
    (cuhe_hip_ct_binop_list, cuhe_hip_crt_add_list) against cuhe_hip_ct_mul / ct_add / crt_add per ciphertext, the array form of
    n2c for non-products (cuhe_hip_intt_batch) against cuhe_hip_ct_intt, the event calls, and the diagnostics
    (allocator counters, generation, transform scratch, dispatch info); round 5: modSwitch and cNot over lists (cuhe_hip_crt_mod_switch_list,
    cuhe_hip_crt_add_int_list), in place and into other blocks, against cuhe_hip_crt_mod_switch / crt_add_int per ciphertext.  A cyclic and a
    negacyclic ring; 70 ciphertexts (two launches of the list kernels)."""
    import ctypes as C
    lib, ck = gu.lib, gu.ck
    g = gu.GpuCtx(*PSETS[name])
    try:
        q = g.prm
        lvl, n = 1, 70
        npr, logq, ctlen = g.np_(lvl), g.logq(lvl), g.ctlen
        rng = np.random.default_rng(77)
        crt = [np.zeros((npr, q.crtLen), dtype=np.uint32) for _ in range(2 * n)]
        for a in crt:
            for t in range(npr):
                a[t, :q.modLen] = rng.integers(0, g.primes[t], q.modLen, dtype=np.uint32)
        dcrt = [gu.to_dev(a) for a in crt]
        dntt = [gu.empty_u64(npr, ctlen) for _ in range(2 * n)]
        for x, X in zip(dcrt, dntt):
            ck(lib.cuhe_hip_ct_ntt(X.data_ptr(), x.data_ptr(), logq, 0, None))
        plist = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        # ---- gather / scatter / copy of blocks
        cbytes = npr * q.crtLen * 4
        arr = gu.empty_u32(n, npr, q.crtLen)
        ck(lib.cuhe_hip_gather_blocks(arr.data_ptr(), plist(dcrt[:n]), n, cbytes, 0, None))
        assert np.array_equal(gu.host_u32(arr), np.stack(crt[:n]))
        outs = [gu.empty_u32(npr, q.crtLen) for _ in range(n)]
        ck(lib.cuhe_hip_scatter_blocks(plist(outs), arr.data_ptr(), n, cbytes, 0, None))
        assert all(np.array_equal(gu.host_u32(o), c) for o, c in zip(outs, crt[:n]))
        outs2 = [gu.empty_u32(npr, q.crtLen) for _ in range(n)]
        ck(lib.cuhe_hip_copy_list(plist(outs2), plist(dcrt[n:]), n, cbytes, 0, None))
        assert all(np.array_equal(gu.host_u32(o), c) for o, c in zip(outs2, crt[n:]))
        assert lib.cuhe_hip_gather_blocks(arr.data_ptr(), plist(dcrt[:n]), n, cbytes + 4, 0, None) != 0      # not a multiple of 16
        # ---- cAnd / cXor over lists against the single-ciphertext entry points
        for mul, single in ((1, lib.cuhe_hip_ct_mul), (0, lib.cuhe_hip_ct_add)):
            z = [gu.empty_u64(npr, ctlen) for _ in range(n)]
            ck(lib.cuhe_hip_ct_binop_list(mul, plist(z), plist(dntt[:n]), plist(dntt[n:]), n, logq, 0, None))
            want = gu.empty_u64(npr, ctlen)
            for i in (0, 17, 63, 64, 69):
                ck(single(want.data_ptr(), dntt[i].data_ptr(), dntt[n + i].data_ptr(), logq, 0, None))
                assert np.array_equal(gu.host_u64(z[i]), gu.host_u64(want)), (mul, i)
        zc = [gu.empty_u32(npr, q.crtLen) for _ in range(n)]
        ck(lib.cuhe_hip_crt_add_list(plist(zc), plist(dcrt[:n]), plist(dcrt[n:]), n, logq, 0, None))
        wantc = gu.empty_u32(npr, q.crtLen)
        for i in (0, 63, 64, 69):
            ck(lib.cuhe_hip_crt_add(wantc.data_ptr(), dcrt[i].data_ptr(), dcrt[n + i].data_ptr(), logq, 0, None))
            assert np.array_equal(gu.host_u32(zc[i]), gu.host_u32(wantc)), i
        # in place (out = first operand), as cXor(out, out, term) records it
        ck(lib.cuhe_hip_crt_add_list(plist(zc), plist(zc), plist(dcrt[n:]), n, logq, 0, None))
        ck(lib.cuhe_hip_crt_add(wantc.data_ptr(), wantc.data_ptr(), dcrt[n + 69].data_ptr(), logq, 0, None))
        assert np.array_equal(gu.host_u32(zc[69]), gu.host_u32(wantc))
        # ---- cNot over a list: into other blocks (every coefficient copied) and in place (constant terms only)
        a_not = q.modMsg - 1
        zn = [gu.empty_u32(npr, q.crtLen) for _ in range(n)]
        for t in zn: t.zero_()
        ck(lib.cuhe_hip_crt_add_int_list(plist(zn), plist(dcrt[:n]), a_not, n, logq, 0, None))
        for i in (0, 63, 64, 69):
            wn = gu.to_dev(crt[i])
            ck(lib.cuhe_hip_crt_add_int(wn.data_ptr(), wn.data_ptr(), a_not, logq, 0, None))
            assert np.array_equal(gu.host_u32(zn[i]), gu.host_u32(wn)), i
            assert np.array_equal(gu.host_u32(zn[i])[:, 0], (crt[i][:, 0].astype(np.uint64) + a_not) % g.primes[:npr].astype(np.uint64)), i
        ck(lib.cuhe_hip_crt_add_int_list(plist(zn), plist(zn), a_not, n, logq, 0, None))            # in place: + 2 (modMsg - 1) in all
        for i in (0, 64, 69):
            want2 = crt[i].copy(); want2[:, 0] = (crt[i][:, 0].astype(np.uint64) + 2 * a_not) % g.primes[:npr].astype(np.uint64)
            assert np.array_equal(gu.host_u32(zn[i]), want2), i
        # ---- modSwitch over a list: into other blocks and inside the ciphertexts' own blocks, against the single entry point
        if npr >= 2 and lvl + 1 < q.depth:
            zm = [gu.empty_u32(npr, q.crtLen) for _ in range(n)]
            for t in zm: t.zero_()
            ck(lib.cuhe_hip_crt_mod_switch_list(plist(zm), plist(dcrt[:n]), lvl, n, 0, None))
            wm = gu.empty_u32(npr, q.crtLen); wm.zero_()
            for i in (0, 17, 63, 64, 69):
                ck(lib.cuhe_hip_crt_mod_switch(wm.data_ptr(), dcrt[i].data_ptr(), logq, 0, None))
                assert np.array_equal(gu.host_u32(zm[i])[:npr - 1], gu.host_u32(wm)[:npr - 1]), i
            own = [gu.to_dev(c) for c in crt[:n]]
            ck(lib.cuhe_hip_crt_mod_switch_list(plist(own), plist(own), lvl, n, 0, None))
            for i in range(n):
                assert np.array_equal(gu.host_u32(own[i])[:npr - 1], gu.host_u32(zm[i])[:npr - 1]), i
                assert np.array_equal(gu.host_u32(own[i])[npr - 1], crt[i][npr - 1]), i                 # the dropped row is never written
            assert lib.cuhe_hip_crt_mod_switch_list(plist(own), plist(own), q.depth - 1, n, 0, None) != 0
        # ---- n2c of non-products over an array
        nbytes = npr * ctlen * 8
        narr = gu.empty_u64(n, npr, ctlen)
        ck(lib.cuhe_hip_gather_blocks(narr.data_ptr(), plist(dntt[:n]), n, nbytes, 0, None))
        back = gu.empty_u32(n, npr, q.crtLen)
        ck(lib.cuhe_hip_intt_batch(back.data_ptr(), narr.data_ptr(), lvl, n, 0, None))
        got = gu.host_u32(back)
        for i in range(n):
            assert np.array_equal(got[i][:, :q.modLen], crt[i][:, :q.modLen]), i          # the transform round trip, every ciphertext
        # ---- events: a second stream waits for the first
        s1, s2, ev = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ck(lib.cuhe_hip_stream_create(0, C.byref(s1))); ck(lib.cuhe_hip_stream_create(0, C.byref(s2))); ck(lib.cuhe_hip_event_create(0, C.byref(ev)))
        a1, a2 = gu.empty_u64(npr, ctlen), gu.empty_u64(npr, ctlen)
        ck(lib.cuhe_hip_ct_mul(a1.data_ptr(), dntt[0].data_ptr(), dntt[1].data_ptr(), logq, 0, s1))
        ck(lib.cuhe_hip_event_record(0, ev, s1))
        ck(lib.cuhe_hip_stream_wait_event(0, s2, ev))
        ck(lib.cuhe_hip_ct_add(a2.data_ptr(), a1.data_ptr(), dntt[2].data_ptr(), logq, 0, s2))
        ck(lib.cuhe_hip_stream_sync(0, s2))
        ck(lib.cuhe_hip_event_sync(0, ev))
        assert lib.cuhe_hip_event_query(0, ev) == 0
        w = gu.empty_u64(npr, ctlen)
        ck(lib.cuhe_hip_ct_mul(w.data_ptr(), dntt[0].data_ptr(), dntt[1].data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_add(w.data_ptr(), w.data_ptr(), dntt[2].data_ptr(), logq, 0, None))
        assert np.array_equal(gu.host_u64(a2), gu.host_u64(w))
        ck(lib.cuhe_hip_event_destroy(0, ev)); ck(lib.cuhe_hip_stream_destroy(0, s1)); ck(lib.cuhe_hip_stream_destroy(0, s2))
        # ---- diagnostics
        cnt = (C.c_longlong * 4)()
        ck(lib.cuhe_hip_alloc_counters(cnt))
        assert cnt[0] >= 0 and cnt[1] >= 0
        gen0 = lib.cuhe_hip_generation()
        info = C.create_string_buffer(256)
        ck(lib.cuhe_hip_last_dispatch_info(0, info, 256))
        assert b"rows of" in info.value
        assert lib.cuhe_hip_ntt_swap(0)
        cinfo = C.create_string_buffer(512)
        ck(lib.cuhe_hip_comm_info(cinfo, 512))
        assert b"rccl" in cinfo.value.lower()
    finally:
        g.close()
    assert lib.cuhe_hip_generation() == gen0 + 1                  # close() shut the library down
