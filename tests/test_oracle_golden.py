"""Pins the C oracle (oracle/) against the pure-Python big-int fixtures in
tests/golden/ and against the reference's own by-definition transform test
(tests/test_ntt.cu:38-64) and constants (cuhe/Base.cu:65,489,656,841)."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_field_vectors(golden):
    g = golden("field.json")
    L = O.lib()
    assert int(g["P"]) == O.P and int(g["g"]) == O.G
    for c in g["cases"]:
        x, y = int(c["x"]), int(c["y"])
        assert L.orc_add_modP(x, y) == int(c["add"])
        assert L.orc_sub_modP(x, y) == int(c["sub"])
        assert L.orc_mul_modP(x, y) == int(c["mul"])
    for s in g["shifts"]:      # tests/test_ModP.cu:45-49: shifts 3*a*b
        assert L.orc_ls_modP(int(s["x"]), s["l"]) == int(s["out"])
    k = g["consts"]
    assert L.orc_pow_modP(O.G, 65536) == int(k["g_pow_65536"]) == 1
    assert L.orc_pow_modP(O.G, 32768) == int(k["g_pow_32768"]) == O.P - 1
    assert L.orc_pow_modP(O.G, 1024) == int(k["g_pow_1024"]) == 8
    # cuhe/Base.cu:489,656,841
    assert L.orc_len_inv(16384) == int(k["inv_16384"]) == 18445618169508003841
    assert L.orc_len_inv(32768) == int(k["inv_32768"]) == 18446181119461294081
    assert L.orc_len_inv(65536) == int(k["inv_65536"]) == 18446462594437939201


@pytest.mark.parametrize("length", [16384, 32768, 65536])
def test_ntt_vectors(golden, length):
    g = golden("ntt.json")[str(length)]
    x = O.splitmix_u32_below(length // 2, g["bound"], g["seed"])
    X = O.ntt_ext(x, length)
    for i, v in g["by_definition"].items():
        assert int(X[int(i)]) == int(v)
    assert [str(int(v)) for v in X[:8]] == g["head"]
    assert sha(X) == g["sha256_full"]
    x2 = O.splitmix_u32_below(length // 2, 0xFFFFFFFF, g["seed32"])
    assert sha(O.ntt_ext(x2, length)) == g["sha256_full32"]
    back = O.intt_modp(X, length, g["intt_prime"])
    assert sha(back) == g["intt_sha256"]
    assert np.array_equal(back[:length // 2], x % g["intt_prime"]) and not back[length // 2:].any()


def test_ntt_fast_equals_definition_16k():
    """tests/test_ntt.cu:38-64 on every output index (the reference checks one slab)."""
    x = O.splitmix_u32_below(8192, 1 << 31, 7)
    assert np.array_equal(O.ntt_naive(x, 16384), O.ntt_ext(x, 16384))


@pytest.mark.parametrize("name", ["dhs_simple", "prince", "toy1155", "pow2_16384", "c3_65536", "c4_65536", "c1_pow2_1prime", "c1_prime_m_1prime"])
def test_params_and_primes(golden, name):
    g = golden("params.json")[name]
    q = O.set_param(*g["args"])
    for k, v in g["params"].items():
        assert getattr(q, k) == v, k
    primes = O.gen_crt_primes(q)
    assert [int(p) for p in primes] == g["primes"]
    L = O.lib()
    import ctypes as C
    for lvl, rec in g["levels"].items():
        assert L.orc_log_coeff(C.byref(q), int(lvl)) == rec["logCoeff"]
        assert L.orc_words_coeff(C.byref(q), int(lvl)) == rec["wordsCoeff"]
        if "numEvalKey" in rec and q.logRelin:
            assert L.orc_num_eval_key(C.byref(q), int(lvl)) == rec["numEvalKey"]
            assert L.orc_num_crt_prime(C.byref(q), int(lvl)) == rec["numCrtPrime"]


def test_survey_prime_sets(golden):
    """worked parameter sets."""
    g = golden("params.json")
    assert g["dhs_simple"]["primes"] == [2097143, 2097133, 524287, 1048573, 1048571, 1048559, 1048549]
    assert g["prince"]["primes"][:3] == [33554393, 33554383, 33554371]
    assert g["prince"]["params"]["numEvalKey"] == 40 and g["prince"]["params"]["nttLen"] == 32768


@pytest.mark.parametrize("name", ["dhs_simple", "toy1155", "pow2_16384"])
def test_ctx_constants(golden, name):
    g = golden("params.json")[name]
    c = O.Ctx(*g["args"])
    for lvl, hx in enumerate(g["coeff_moduli"]):
        assert c.coeff_modulus(lvl) == int(hx, 16)
    invp = c.invp()
    assert hashlib.sha256(invp.tobytes()).hexdigest() == g["invp_sha256"]
    assert [int(v) for v in invp[:6]] == g["invp_head"]
    c.close()


def test_cyclotomic():
    assert list(O.cyclotomic(8191)) == [1] * 8191
    c = O.cyclotomic(16384)
    assert c[0] == 1 and c[8192] == 1 and not c[1:8192].any()
    assert list(O.cyclotomic(15)) == [1, -1, 0, 1, -1, 1, 0, -1, 1]
    assert len(O.cyclotomic(21845)) == 16385
    assert list(O.cyclotomic(105))[7] == -2      # first cyclotomic with a coefficient of magnitude 2


def _pipeline(golden, name, fixture):
    g = golden(fixture)
    c = O.Ctx(*g["args"])
    q = c.prm
    for lvl_s, rec in g["levels"].items():
        lvl = int(lvl_s)
        W, M = c.words(lvl), c.coeff_modulus(lvl)
        assert W == rec["words"] and c.np_(lvl) == rec["num_primes"]
        a_raw, a = O.random_raw(q.rawLen, q.modLen, W, M, rec["seed_a"])
        b_raw, b = O.random_raw(q.rawLen, q.modLen, W, M, rec["seed_b"])
        crt_a = c.crt(a_raw, lvl)
        assert sha(crt_a) == rec["crt_a_sha256"]
        # ICRT round trip
        assert np.array_equal(c.icrt(crt_a, lvl), a_raw)
        out = c.mul_raw(a_raw, b_raw, lvl)
        assert sha(out) == rec["mul_sha256"]
        if "mul_hex" in rec:
            assert [hex(v) for v in O.raw_to_ints(out, q.modLen)] == rec["mul_hex"]
        crt_b = c.crt(b_raw, lvl)
        assert sha(c.crt_add(crt_a, crt_b)) == rec["crt_add_sha256"]
        if "modswitch_sha256" in rec:
            assert sha(c.modswitch(crt_a)) == rec["modswitch_sha256"]
    if "relin" in g:
        K, W0 = q.numEvalKey, c.words(0)
        M0 = c.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(q.rawLen, q.modLen, W0, M0, g["relin"]["seed_ek_base"] + j)[0]
                           for j in range(K)])
        ek = c.init_relin(ek_raw)
        for lvl_s, rec in g["relin"]["levels"].items():
            lvl = int(lvl_s)
            ct_raw, _ = O.random_raw(q.rawLen, q.modLen, c.words(lvl), c.coeff_modulus(lvl), rec["seed_ct"])
            assert c.nkeys(lvl) == rec["num_keys"]
            res = c.intt_mod(c.relin(ct_raw, lvl, ek))
            assert sha(res) == rec["crt_sha256"]
            if rec.get("head"):
                assert [[int(v) for v in row[:4]] for row in res] == rec["head"]
    c.close()


def test_pipeline_toy(golden):
    _pipeline(golden, "toy1155", "pipeline_toy1155.json")


def test_pipeline_pow2(golden):
    _pipeline(golden, "pow2_16384", "pipeline_pow2_16384.json")


@pytest.mark.parametrize("name", ["c1_pow2_1prime", "c1_prime_m_1prime"])
def test_pipeline_config1_single_prime(golden, name):
    """BASELINE config 1 exactly: N = 2^13, ONE CRT prime (p = 2097143), host only: the polynomial product on
    x^8192 + 1 and on Phi_8191 against the pure-Python fixture (examples/DHS/DHS.cu:219-221 meaning)"""
    g = golden("params.json")[name]
    assert g["primes"] == [2097143] and g["params"]["numCrtPrime"] == 1 and g["params"]["modLen2"] == 8192
    _pipeline(golden, name, "pipeline_%s.json" % name)


def test_pipeline_dhs(golden):
    _pipeline(golden, "dhs_simple", "pipeline_dhs_simple.json")


@pytest.mark.parametrize("args", [(3, 2, 8, 40, 20, 1155), (5, 2, 1, 61, 20, 8191), (3, 2, 16, 50, 25, 16384)])
def test_barrett_restatement_equals_exact_remainder(args):
    """cuhe/Operations.cu:460-501 (Barrett through NTTs) == exact remainder mod Phi_m."""
    c = O.Ctx(*args)
    q = c.prm
    np_ = q.numCrtPrime
    rng = np.random.default_rng(5)
    a = np.zeros((np_, q.crtLen), dtype=np.uint32); b = np.zeros_like(a)
    for i, p in enumerate(c.primes):
        a[i, :q.modLen] = rng.integers(0, p, q.modLen); b[i, :q.modLen] = rng.integers(0, p, q.modLen)
    hold = c.intt_hold(c.ntt_mul(c.ntt(a), c.ntt(b)))
    assert np.array_equal(c.barrett(hold), c.poly_reduce_exact(hold))
    # worst-case magnitudes: all residues p-1
    for i, p in enumerate(c.primes):
        a[i, :q.modLen] = p - 1
    hold = c.intt_hold(c.ntt_mul(c.ntt(a), c.ntt(a)))
    assert np.array_equal(c.barrett(hold), c.poly_reduce_exact(hold))
    c.close()


def test_fold_reduction_equals_division_and_python_ints():
    """Round 6: the oracle reduces 128-bit values with the fold 2^64 = 2^32 - 1, 2^96 = -1 (mod P) the reference's field arithmetic uses
    (cuhe/ModP.h:249-289) instead of `u128 % P`.  Checked against the division forms kept for this purpose and against Python integers on
    random operands, on every pair of edge values (including unreduced ones: P, P + 1, 2^64 - 1) and on the products that exercise each
    branch of the fold (borrow of hh, carry of the middle term, result in [P, 2^64))."""
    import random
    import oracle_lib as O
    L = O.lib()
    P = O.P
    edge = [0, 1, 2, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, P - 2, P - 1, P, P + 1, (1 << 64) - 2, (1 << 64) - 1, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF,
            0xFFFFFFFE00000001, 0xFFFFFFFEFFFFFFFF, 1 << 63, (1 << 63) + 1]
    pairs = [(x, y) for x in edge for y in edge]
    rnd = random.Random(6)
    pairs += [(rnd.getrandbits(64), rnd.getrandbits(64)) for _ in range(20000)]
    pairs += [(rnd.getrandbits(64), rnd.getrandbits(k)) for k in (1, 31, 32, 33) for _ in range(2000)]
    for x, y in pairs:
        assert L.orc_mul_modP(x, y) == L.orc_mul_modP_div(x, y) == x * y % P, (x, y)
        assert L.orc_add_modP(x, y) == L.orc_add_modP_div(x, y) == (x + y) % P, (x, y)
        assert L.orc_sub_modP(x, y) == (x - y) % P, (x, y)


@pytest.mark.parametrize("length", [16384, 32768, 65536])
def test_throughput_form_of_the_transform_equals_the_oracle_transform(length):
    """orc_ntt_ext_fast_batch (tables shared by the batch; what bench.py times as cpu_baseline) against orc_ntt_ext on random rows, all-ones,
    zero and unit rows, one and several threads."""
    import oracle_lib as O
    rng = np.random.default_rng(length)
    x = rng.integers(0, 1 << 32, (5, length // 2), dtype=np.uint64).astype(np.uint32)
    x[1] = 0xFFFFFFFF; x[2] = 0; x[3] = 0; x[3, 0] = 1
    want = np.stack([O.ntt_ext(r, length) for r in x])
    for threads in (1, 4):
        got, used = O.ntt_ext_fast_batch(x, length, threads)
        assert used >= 1 and np.array_equal(got, want)
    assert (got[3] == 1).all() and not got[2].any()


def test_host_cores_respects_the_cgroup_quota(tmp_path, monkeypatch):
    """the thread count of the CPU baselines is what the process may really use: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs under a quota of 16; 128 OpenMP threads run 2x slower there than 16: profiles/r06_cpu_thread_scaling.txt)"""
    import builtins
    import os
    import oracle_lib as O
    real_open = builtins.open

    def fake(content):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                f = tmp_path / "cpu.max"
                f.write_text(content)
                return real_open(f, *a, **k)
            return real_open(path, *a, **k)
        return _open
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert O.host_cores() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert O.host_cores() == 256
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert O.host_cores() == 1
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert O.host_cores() == 8
