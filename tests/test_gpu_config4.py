"""-m gpu: BASELINE config 4 at FULL size (48 CRT primes, 72 / 69 evaluation keys of 16-bit windows) pinned by exact integers
that do not come from the oracle:

 * the batched key switch on the matrix cores (cuhe_hip_relin_batch -> k_relin_mac_mfma<1,32>: three 16-prime tiles, a
   full ciphertext tile and a partial one) for 21 DISTINCT ciphertexts, every ciphertext, every prime, every coefficient;
 * the CRT-prime-sharded multiply + relinearise (cuhe_hip_mul_relin_sharded_inproc) over 2, 4, 5 and 8 virtual devices
   (5: unequal blocks), every device holding the keys of its own primes only (cuhe_hip_init_relin_sharded);
 * the 2-rank launch of bench.py (gloo, both ranks on the one GPU): its N > 1 legs run and report no error.

With evaluation keys ek_j = s * 2^(w j) mod q0 the key-switch sum  sum_j window_j(c) * ek_j  equals  c * s  modulo x^n + 1 and
q (the windows recompose c); q is the product of the level's primes, so the residue of the result modulo a prime p is
(c * s mod x^n + 1) mod p -- for a sparse s a few negacyclic shifts of the residue rows, exact in 64-bit numpy arithmetic.
Reference: cuhe/Relinearization.cu:43-88, cuhe/CuHE.cu:101,570-581 (the chain), cuhe/CuHE.cu:217-256 (devices)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RINGS = {"x^32768+1": (25, 2, 16, 576, 24, 65536), "x^65536+1": (25, 2, 16, 552, 23, 131072)}


@pytest.fixture(scope="module")
def gu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X (no CPU fallback exists for the HIP path)")
    import gpu_util
    return gpu_util


def sparse_terms(n, q0, seed):
    rng = np.random.default_rng(seed)
    return [(int(e), int.from_bytes(rng.bytes(150), "little") % q0) for e in (0, 1, 777, 20011, n - 1)]          # s = sum v x^e


def structured_keys(terms, K, w, q0, raw_len, W0):
    ek = np.zeros((K, raw_len, W0), dtype=np.uint32)
    for j in range(K):
        for e, v in terms:
            ek[j, e] = np.frombuffer(((v << (w * j)) % q0).to_bytes(4 * W0, "little"), dtype=np.uint32)
    return ek


def negacyclic_times_sparse(rows, terms, primes, n):
    """rows u32[np][>= n] residues of c; returns the residues of c * s mod (x^n + 1), s = sum v x^e, as u64[np][n]"""
    out = np.zeros((len(primes), n), dtype=np.uint64)
    for t, p in enumerate(primes):
        c = rows[t, :n].astype(np.uint64)
        acc = np.zeros(n, dtype=np.uint64)
        for e, v in terms:
            vp = np.uint64(v % p)
            sh = np.concatenate(((np.uint64(p) - c[n - e:]) % np.uint64(p), c[:n - e])) if e else c      # x^e c: the wrapped part changes sign
            acc = (acc + sh * vp) % np.uint64(p)                                                            # < 2^24 * 2^24 + 2^24
        out[t] = acc
    return out


@pytest.mark.parametrize("ring", list(RINGS))
def test_batched_key_switch_21_distinct_ciphertexts_vs_integers(gu, ring):
    lib, ck = gu.lib, gu.ck
    g = gu.GpuCtx(*RINGS[ring])
    try:
        q = g.prm
        n, K, W0, q0 = q.modLen, q.numEvalKey, g.words(0), g.coeff_modulus(0)
        assert q.numCrtPrime == 48 and K in (72, 69)
        primes = g.crt_primes()
        terms = sparse_terms(n, q0, 11)
        g.init_relin(structured_keys(terms, K, q.logRelin, q0, q.rawLen, W0))
        B = 21
        for lvl in (0, 3):
            npr = g.np_(lvl)
            rng = np.random.default_rng(100 + lvl)
            src = np.zeros((B, npr, q.crtLen), dtype=np.uint32)
            for t in range(npr):
                src[:, t, :n] = rng.integers(0, primes[t], (B, n), dtype=np.uint32)
            src[0, :, :n] = 0                                   # edge rows: zero, all p - 1, a single non-zero coefficient
            for t in range(npr):
                src[1, t, :n] = primes[t] - 1
            src[2, :, :n] = 0; src[2, :, n - 1] = 1
            d_src = gu.to_dev(src.reshape(B * npr, q.crtLen))
            d_dst = gu.empty_u32(B * npr, q.crtLen)
            ck(lib.cuhe_hip_set_relin_mfma(5))                  # the matrix-core kernel from 5 ciphertexts on (the default)
            ck(lib.cuhe_hip_relin_batch(d_dst.data_ptr(), d_src.data_ptr(), lvl, B, 0, None))
            got = gu.host_u32(d_dst).reshape(B, npr, q.crtLen)
            for b in range(B):
                want = negacyclic_times_sparse(src[b], terms, primes[:npr], n)
                assert np.array_equal(got[b, :, :n].astype(np.uint64), want), (ring, lvl, b)
            # the same rows through the VALU kernels (matrix cores off): bit-identical
            ck(lib.cuhe_hip_set_relin_mfma(0))
            d_dst2 = gu.empty_u32(B * npr, q.crtLen)
            ck(lib.cuhe_hip_relin_batch(d_dst2.data_ptr(), d_src.data_ptr(), lvl, B, 0, None))
            assert np.array_equal(gu.host_u32(d_dst2).reshape(B, npr, q.crtLen), got), (ring, lvl)
            ck(lib.cuhe_hip_set_relin_mfma(5))
            # and with the inverse CRT on the VALU kernel instead of the matrix cores (cuhe_hip_set_icrt_mfma): bit-identical
            ck(lib.cuhe_hip_set_icrt_mfma(0))
            ck(lib.cuhe_hip_relin_batch(d_dst2.data_ptr(), d_src.data_ptr(), lvl, B, 0, None))
            assert np.array_equal(gu.host_u32(d_dst2).reshape(B, npr, q.crtLen), got), (ring, lvl)
            ck(lib.cuhe_hip_set_icrt_mfma(1))
    finally:
        lib.cuhe_hip_set_relin_mfma(5)
        lib.cuhe_hip_set_icrt_mfma(1)
        g.close()


def test_sharded_multiply_at_config4_size_on_2_4_5_8_virtual_devices(gu):
    """a * b with a sparse b (so that the product is exact in numpy), then the key switch: residues of a * b * s."""
    lib, ck = gu.lib, gu.ck
    args = RINGS["x^32768+1"]
    try:
        for ndev in (2, 4, 5, 8):
            lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
            ck(lib.cuhe_hip_set_virtual_devices(1))
            ck(lib.cuhe_hip_set_parameters(*args))
            ck(lib.cuhe_hip_multi_gpus(ndev))
            ck(lib.cuhe_hip_init(None, 0))
            q = gu.capi.get_params()
            n, K = q.modLen, q.numEvalKey
            W0 = lib.cuhe_hip_words_coeff(0)
            buf = (ctypes.c_uint8 * 4096)(); ln = ctypes.c_size_t(0)
            ck(lib.cuhe_hip_get_coeff_modulus(0, buf, 4096, ctypes.byref(ln)))
            q0 = int.from_bytes(bytes(buf[:ln.value]), "little")
            pr = (ctypes.c_uint32 * q.numCrtPrime)()
            ck(lib.cuhe_hip_get_crt_primes(pr, q.numCrtPrime))
            primes = [int(x) for x in pr]
            terms = sparse_terms(n, q0, 12)
            ek = np.ascontiguousarray(structured_keys(terms, K, q.logRelin, q0, q.rawLen, W0))
            ck(lib.cuhe_hip_init_relin_sharded(ek.ctypes.data_as(ctypes.c_void_p)))        # device d: the keys of its primes only
            f, c = ctypes.c_int(), ctypes.c_int()
            held = 0
            for d in range(ndev):
                ck(lib.cuhe_hip_key_range(ndev, d, ctypes.byref(f), ctypes.byref(c)))
                held += c.value
            assert held <= q.numCrtPrime + (q.depth - 1) * (ndev - 1), (ndev, held)          # ~1/ndev each (blocks overlap by the level drift only)
            bterms = [(0, 3), (5, 1), (n - 2, 7)]
            ctlen = lib.cuhe_hip_ct_len()
            for lvl in (0, 4):
                npr, logq = lib.cuhe_hip_num_crt_prime(lvl), lib.cuhe_hip_log_coeff(lvl)
                rng = np.random.default_rng(200 + lvl + ndev)
                a = np.zeros((npr, q.crtLen), dtype=np.uint32)
                b = np.zeros((npr, q.crtLen), dtype=np.uint32)
                for t in range(npr):
                    a[t, :n] = rng.integers(0, primes[t], n, dtype=np.uint32)
                    for e, v in bterms:
                        b[t, e] = v
                ab = negacyclic_times_sparse(a, bterms, primes[:npr], n)
                want = negacyclic_times_sparse(ab.astype(np.uint32), terms, primes[:npr], n)
                for dev0 in (0, ndev - 1):
                    na, nb = gu.empty_u64(npr, ctlen), gu.empty_u64(npr, ctlen)
                    ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), gu.to_dev(a).data_ptr(), logq, dev0, None))
                    ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), gu.to_dev(b).data_ptr(), logq, dev0, None))
                    out = gu.empty_u32(npr, q.crtLen)
                    for rep in range(2):
                        out.zero_()
                        ck(lib.cuhe_hip_mul_relin_sharded_inproc(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, dev0, None))
                        ck(lib.cuhe_hip_stream_sync(dev0, None))
                        assert np.array_equal(gu.host_u32(out)[:, :n].astype(np.uint64), want), (ndev, lvl, dev0, rep)
            # an entry point that needs EVERY prime's keys says so on a device that holds a part of them
            raw = gu.empty_u32(q.rawLen, W0)
            acc = gu.empty_u64(q.numCrtPrime, ctlen)
            assert lib.cuhe_hip_relinearization(acc.data_ptr(), raw.data_ptr(), 0, 0, None) != 0
            assert b"holds" in lib.cuhe_hip_last_error()
    finally:
        lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters(); lib.cuhe_hip_set_virtual_devices(0); lib.cuhe_hip_multi_gpus(1)


def test_sharded_multiply_through_rccl_on_one_rank_at_config4_size(gu):
    """cuhe_hip_mul_relin_sharded (the one-process-per-GPU form) at the FULL config-4 size through a communicator of one rank
    with the exchange FORCED (cuhe_hip_comm_force_exchange): librccl is opened, the communicator made, and the collective on
    the CRT rows really runs on the compute stream between the two stages -- everything the N > 1 path does except a second
    GPU.  All three forms of the exchange run: the in-place ncclAllGather the policy picks for equal blocks, the padded
    ncclAllGather (staging buffer + strided unpack) and the broadcast group.  Result against exact integers (a * b * s with
    sparse b and s), two levels, twice per buffer set; cuhe_hip_comm_info must report what RCCL says about the communicator and
    which paths were taken."""
    lib, ck = gu.lib, gu.ck
    g = gu.GpuCtx(*RINGS["x^32768+1"])
    try:
        q = g.prm
        n, K, W0, q0 = q.modLen, q.numEvalKey, g.words(0), g.coeff_modulus(0)
        primes = g.crt_primes()
        terms = sparse_terms(n, q0, 13)
        g.init_relin(structured_keys(terms, K, q.logRelin, q0, q.rawLen, W0))
        uid = (ctypes.c_uint8 * 128)()
        ck(lib.cuhe_hip_comm_unique_id(uid))
        ck(lib.cuhe_hip_comm_init(1, 0, uid))
        ck(lib.cuhe_hip_comm_force_exchange(1))
        info = ctypes.create_string_buffer(512)
        bterms = [(0, 3), (5, 1), (n - 2, 7)]
        ctlen = lib.cuhe_hip_ct_len()
        for lvl in (0, 4):
            npr, logq = g.np_(lvl), g.logq(lvl)
            rng = np.random.default_rng(300 + lvl)
            a = np.zeros((npr, q.crtLen), dtype=np.uint32)
            b = np.zeros((npr, q.crtLen), dtype=np.uint32)
            for t in range(npr):
                a[t, :n] = rng.integers(0, primes[t], n, dtype=np.uint32)
                for e, v in bterms:
                    b[t, e] = v
            want = negacyclic_times_sparse(negacyclic_times_sparse(a, bterms, primes[:npr], n).astype(np.uint32), terms, primes[:npr], n)
            na, nb = gu.empty_u64(npr, ctlen), gu.empty_u64(npr, ctlen)
            ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), gu.to_dev(a).data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), gu.to_dev(b).data_ptr(), logq, 0, None))
            out = gu.empty_u32(npr, q.crtLen)
            for force in (1, 2, 3):                        # the policy (one rank: equal blocks -> in place), padded, broadcast group
                ck(lib.cuhe_hip_comm_force_exchange(force))
                assert lib.cuhe_hip_exchange_path(lvl, 1, force) == force
                for rep in range(2):
                    out.zero_()
                    ck(lib.cuhe_hip_mul_relin_sharded(out.data_ptr(), na.data_ptr(), nb.data_ptr(), lvl, 0, None))
                    ck(lib.cuhe_hip_stream_sync(0, None))
                    assert np.array_equal(gu.host_u32(out)[:, :n].astype(np.uint64), want), (lvl, force, rep)
        ck(lib.cuhe_hip_comm_info(info, 512))
        txt = info.value.decode()
        assert "ncclCommCount 1" in txt and "ncclCommUserRank 0" in txt and "exchanges so far 12" in txt, txt
        assert "ncclAllGather in place 4, padded 4, broadcast group 4" in txt and "ncclBroadcast" in txt, txt
    finally:
        lib.cuhe_hip_comm_destroy()
        g.close()


def test_bench_two_ranks_on_one_gpu_runs_every_leg():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --dist-backend gloo with both ranks on GPU 0:
    the N > 1 legs (sharded multiply with its exchange, replicated multiplies, PRINCE over two devices) complete."""
    env = dict(os.environ, CUHE_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CUHE_BENCH_WATCHDOG="240")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--no-cpu", "--batch", "1024"]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=330)
    except subprocess.TimeoutExpired as ex:        # a rank that stops answering dumps its stacks after 240 s (CUHE_BENCH_WATCHDOG)
        pytest.fail("2-rank bench did not finish: " + str(ex.stderr or b"")[-3000:])
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2
    for leg in ("mul_relin_sharded", "mul_relin_replicated", "prince"):
        assert d.get(leg) and "error" not in d[leg], (leg, d.get(leg))
    sh = d["mul_relin_sharded"]
    assert sh["primes_per_rank"] == 24 and sh["key_primes_per_rank"] <= 30 and 0 < sh["serial_fraction"] < 1
    assert d["prince"]["known_answer_ok"] is True


def dense_config4_case(args, B, seed):
    """inputs of the dense-key test and what the ORACLE makes of them (CPU only; timed here so that the GPU test's budget is
    known): dense random evaluation keys (every word random, below q0), B distinct operand pairs with uniformly random
    residues, the oracle's cAnd + relin of every pair (orc_nc_mul_relin_crt_batch: the reference's chain CuHE.cu:101,570-581
    per prime through the negacyclic restatement, OpenMP over the primes)."""
    import oracle_lib as O
    O.set_threads(0)
    o = O.Ctx(*args)
    try:
        q = o.prm
        K, W0, n, npr = q.numEvalKey, o.words(0), q.modLen, o.np_(0)
        rng = np.random.default_rng(seed)
        ek_raw = rng.integers(0, 1 << 32, (K, q.rawLen, W0), dtype=np.uint32)
        ek_raw[:, :, W0 - 1] &= np.uint32((1 << (o.logq(0) - 32 * (W0 - 1) - 1)) - 1)      # every coefficient below 2^(logq0 - 1) < q0
        ek_raw[:, n:, :] = 0
        a = np.stack([np.stack([rng.integers(0, p, q.crtLen, dtype=np.uint32) for p in o.primes[:npr]]) for _ in range(B)])
        b = np.stack([np.stack([rng.integers(0, p, q.crtLen, dtype=np.uint32) for p in o.primes[:npr]]) for _ in range(B)])
        a[0, :, 0] = o.primes[:npr] - 1                                                    # an edge value in every row of the first pair
        want = o.nc_mul_relin_crt_batch(a, b, 0, o.key_residues(ek_raw))
        return ek_raw, a, b, want
    finally:
        o.close()


@pytest.mark.parametrize("ring", list(RINGS))
def test_dense_keys_at_config4_shape_vs_oracle(gu, ring):
    """the BENCHMARKED shape (48 primes, 72 / 69 keys, level 0) with DENSE random keys against the oracle --
    three distinct ciphertext pairs through cuhe_hip_mul_relin_batch on the matrix cores (batch of 5: the three pairs + two
    repeats, the smallest batch that takes k_relin_mac_mfma), through the VALU batch kernels (batch of 3) and through the
    single chain (ct_mul ; ct_intt ; icrt ; relinearization ; ct_intt); every prime, every coefficient."""
    lib, ck = gu.lib, gu.ck
    B = 3
    ek_raw, a, b, want = dense_config4_case(RINGS[ring], B, 0xC4 + len(ring))
    g = gu.GpuCtx(*RINGS[ring])
    try:
        q = g.prm
        npr, ctlen = g.np_(0), lib.cuhe_hip_ct_len()
        assert (q.numCrtPrime, npr) == (48, 48) and q.numEvalKey in (72, 69) and g.nc
        assert [int(p) for p in g.crt_primes()] == [int(p) for p in oracle_primes(RINGS[ring])]
        g.init_relin(ek_raw)
        for t in range(B):                                       # the single chain
            assert np.array_equal(g.mul_relin_crt(a[t], b[t], 0), want[t]), (ring, "single", t)
        for t in range(B):                                       # relinearization ; n2c as the ONE call CuCtxt::relin makes (cuhe_hip_relin_crt)
            assert np.array_equal(g.mul_relin_crt(a[t], b[t], 0, fused=True), want[t]), (ring, "single, one call", t)
        order = [0, 1, 2, 1, 0]                                  # 5 ciphertexts: the matrix-core inner product (default from 5 on)
        na, nb = gu.empty_u64(len(order) * npr, ctlen), gu.empty_u64(len(order) * npr, ctlen)
        for s, t in enumerate(order):
            ck(lib.cuhe_hip_ct_ntt(na[s * npr:].data_ptr(), gu.to_dev(a[t]).data_ptr(), g.logq(0), 0, None))
            ck(lib.cuhe_hip_ct_ntt(nb[s * npr:].data_ptr(), gu.to_dev(b[t]).data_ptr(), g.logq(0), 0, None))
        out = gu.empty_u32(len(order) * npr, q.crtLen)
        ck(lib.cuhe_hip_set_relin_mfma(5))
        ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), 0, len(order), 0, None))
        got = gu.host_u32(out).reshape(len(order), npr, q.crtLen)
        for s, t in enumerate(order):
            assert np.array_equal(got[s], want[t]), (ring, "batch of 5 (matrix cores)", s)
        out.zero_()
        ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), na.data_ptr(), nb.data_ptr(), 0, B, 0, None))          # 3 < 5: the VALU batch kernels
        got = gu.host_u32(out).reshape(len(order), npr, q.crtLen)
        for t in range(B):
            assert np.array_equal(got[t], want[t]), (ring, "batch of 3 (VALU)", t)
    finally:
        lib.cuhe_hip_set_relin_mfma(5)
        g.close()


def oracle_primes(args):
    import oracle_lib as O
    return O.gen_crt_primes(O.set_param(*args))
