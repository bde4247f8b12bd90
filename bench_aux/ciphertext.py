"""BASELINE configs 3 and 4 on one GPU: the full multiply raw -> raw and ciphertext multiply + relinearise (single, batched, concurrent)."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

from .record import HBM_PEAK_GBS, ROOT
from .tables import dispatch_info


def bench_mul_full(lib, ck, torch, np, dev, with_cpu=True, batch=16, cyclic=False):
    """BASELINE config 3: N = 2^15 (64K-point transforms), 32 CRT primes, full multiply of two raw polynomials
    CRT -> NTT -> pointwise -> INTT (+ reduction mod x^n+1) -> ICRT, device resident (mulZZX without the ZZX<->raw staging)."""
    from cuhe_amd import capi
    d, p, w, mn, cut, m = 9, 2, 16, 576, 24, 65536
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    ck(lib.cuhe_hip_set_negacyclic(0 if cyclic else -1))
    ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
    ck(lib.cuhe_hip_init(None, 0))
    q = capi.get_params()
    npn, L, W, logq = q.numCrtPrime, lib.cuhe_hip_ct_len(), lib.cuhe_hip_words_coeff(0), lib.cuhe_hip_log_coeff(0)
    rep = "negacyclic %d-point" % L if lib.cuhe_hip_ct_negacyclic() else "cyclic %d-point" % L
    gen = torch.Generator(device=dev); gen.manual_seed(9)
    ra = torch.randint(-(1 << 31), (1 << 31) - 1, (q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
    rb = torch.randint(-(1 << 31), (1 << 31) - 1, (q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
    ca = torch.zeros((npn, q.crtLen), dtype=torch.int32, device=dev); cb = torch.zeros_like(ca)
    na = torch.empty((npn, L), dtype=torch.int64, device=dev); nb = torch.empty_like(na)
    out = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)

    def one():
        ck(lib.cuhe_hip_crt(ca.data_ptr(), ra.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_crt(cb.data_ptr(), rb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), ca.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), cb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_mul(na.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))
        ck(lib.cuhe_hip_ct_intt(ca.data_ptr(), na.data_ptr(), logq, 1, 0, None))
        ck(lib.cuhe_hip_icrt(out.data_ptr(), ca.data_ptr(), logq, 0, None))

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    got = out.cpu().numpy().view(np.uint32)
    ha, hb = ra.cpu().numpy().view(np.uint32), rb.cpu().numpy().view(np.uint32)
    # ---- the same multiplication for B independent operand pairs per call (cuhe_hip_mul_raw_batch)
    batched = None
    try:
        B = batch
        # B DISTINCT operand pairs; every result row is compared with the single chain on the same pair
        rab = torch.randint(-(1 << 31), (1 << 31) - 1, (B * q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
        rbb = torch.randint(-(1 << 31), (1 << 31) - 1, (B * q.rawLen, W), dtype=torch.int32, device=dev, generator=gen)
        outb = torch.empty((B * q.rawLen, W), dtype=torch.int32, device=dev)
        for _ in range(2):
            ck(lib.cuhe_hip_mul_raw_batch(outb.data_ptr(), rab.data_ptr(), rbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        keep_a, keep_b = ra.clone(), rb.clone()
        for i in range(B):
            ra.copy_(rab[i * q.rawLen:(i + 1) * q.rawLen]); rb.copy_(rbb[i * q.rawLen:(i + 1) * q.rawLen])
            one()
            assert torch.equal(outb[i * q.rawLen:(i + 1) * q.rawLen], out), "batched result %d differs from the single chain" % i
        ra.copy_(keep_a); rb.copy_(keep_b); one(); torch.cuda.synchronize()
        breps = max(3, 64 // B)
        t0 = time.perf_counter()
        for _ in range(breps):
            ck(lib.cuhe_hip_mul_raw_batch(outb.data_ptr(), rab.data_ptr(), rbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        bdt = (time.perf_counter() - t0) / breps / B
        batched = {"value": round(1.0 / bdt, 1), "unit": "full multiplies/s (raw -> raw)", "ms_per_multiply": round(bdt * 1e3, 4), "batch": B,
                   "checked": "%d distinct operand pairs, every result equal to the single chain" % B}
    except Exception as ex:
        batched = {"error": repr(ex)[:300]}
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters(); lib.cuhe_hip_set_negacyclic(-1)
    res = {"value": round(1.0 / dt, 1), "unit": "full multiplies/s (raw -> raw)", "ms": round(dt * 1e3, 4),
           "params": {"setParameters": [d, p, w, mn, cut, m], "numCrtPrime": npn, "transform": rep, "coeff_words": W},
           "transforms_per_multiply": 3 * npn, "batched": batched}
    if with_cpu:
        # the same multiply on the host through the oracle with OpenMP over the CRT primes on all cores (checker + reported
        # CPU baseline, never the product path), and -- when the box has libgmp -- the way the reference's host library does
        # it: ONE big-integer multiplication of Kronecker-packed operands (NTL, which the reference calls at
        # examples/DHS/DHS.cu:219-221, is not installed in this image; it builds on GMP)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        o = O.Ctx(d, p, w, mn, cut, m)
        q0 = o.coeff_modulus(0)
        used = O.set_threads(0)
        o.mul_raw(ha, hb, 0)                              # warm-up (thread pool, page faults)
        t1 = time.perf_counter()
        want = o.mul_raw(ha, hb, 0)
        cdt = time.perf_counter() - t1
        O.set_threads(1)
        o.close()
        assert np.array_equal(got, want), "GPU full multiply differs from the oracle"
        res["cpu_baseline"] = {"value": round(1.0 / cdt, 3), "unit": "full multiplies/s", "cores": used, "kind": "port",
                               "sample": "1 multiply (N=2^15, 32 primes) through oracle/oracle.c, OpenMP over the CRT primes, %.2f s" % cdt}
        t1 = time.perf_counter()
        gm = O.gmp_mul_xn1(ha, hb, q0)
        gdt = time.perf_counter() - t1
        res["cpu_baseline_gmp"] = None
        if gm is not None:
            assert np.array_equal(gm, want), "GMP product differs from the oracle"
            res["cpu_baseline_gmp"] = {"value": round(1.0 / gdt, 3), "unit": "full multiplies/s", "cores": 1, "kind": "port",
                                       "sample": "1 multiply: Kronecker substitution + one mpz_mul + coefficient reduction (libgmp opened at run time; "
                                                 "stand-in for NTL's ZZX multiply, which is not installed), %.2f s" % gdt}
    return res


def bench_mulrelin(lib, ck, torch, np, dev, args, params, with_cpu=False):
    """DHS ciphertext multiply + relinearise per second on 64K-point transforms (BASELINE config 4 shape:
    48 CRT primes < 2^24, w = 16).  NTT-domain operands -> reduced CRT-domain result, keys resident in HBM."""
    d, p, w, mn, cut, m = params
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    ck(lib.cuhe_hip_set_negacyclic(0 if args.cyclic else -1))
    ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
    ck(lib.cuhe_hip_init(None, 0))
    from cuhe_amd import capi
    q = capi.get_params()
    npn, L, K, W = q.numCrtPrime, lib.cuhe_hip_ct_len(), q.numEvalKey, lib.cuhe_hip_words_coeff(0)
    rep = "negacyclic %d-point" % L if lib.cuhe_hip_ct_negacyclic() else "cyclic %d-point" % L
    rng = np.random.default_rng(7)
    ek = rng.integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32)
    ek[:, :, W - 1] &= 0x7FFF                        # keep below 2^(32W-17): any value works, crt reduces
    t0 = time.perf_counter()
    ck(lib.cuhe_hip_init_relin(ek.ctypes.data_as(C.c_void_p)))
    init_s = time.perf_counter() - t0
    logq = lib.cuhe_hip_log_coeff(0)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    a = torch.randint(0, 1 << (q.logCrtPrime - 1), (npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
    b = torch.randint(0, 1 << (q.logCrtPrime - 1), (npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
    na = torch.empty((npn, L), dtype=torch.int64, device=dev); nb = torch.empty_like(na); nc = torch.empty_like(na)
    cr = torch.empty((npn, q.crtLen), dtype=torch.int32, device=dev)
    raw = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)
    ck(lib.cuhe_hip_ct_ntt(na.data_ptr(), a.data_ptr(), logq, 0, None))
    ck(lib.cuhe_hip_ct_ntt(nb.data_ptr(), b.data_ptr(), logq, 0, None))

    fused = True          # what CuCtxt::relin does since round 5; the two-call form is timed beside it below

    def one():
        ck(lib.cuhe_hip_ct_mul(nc.data_ptr(), na.data_ptr(), nb.data_ptr(), logq, 0, None))       # cAnd
        ck(lib.cuhe_hip_ct_intt(cr.data_ptr(), nc.data_ptr(), logq, 1, 0, None))                   # relin: x2r
        ck(lib.cuhe_hip_icrt(raw.data_ptr(), cr.data_ptr(), logq, 0, None))
        if fused:       # relinearization ; n2c as the one call CuCtxt::relin makes (round 5)
            ck(lib.cuhe_hip_relin_crt(cr.data_ptr(), raw.data_ptr(), 0, 0, None))
        else:
            ck(lib.cuhe_hip_relinearization(nc.data_ptr(), raw.data_ptr(), 0, 0, None))
            ck(lib.cuhe_hip_ct_intt(cr.data_ptr(), nc.data_ptr(), logq, 1, 0, None))                   # n2c

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    first_rows = cr.cpu().numpy().view(np.uint32)     # the rows of (a, b): checked against the oracle chain by the CPU leg below
    single_dispatch = dispatch_info(lib)              # (of the chain's last transform call: the inverse rows of the result)
    # the same chain with relinearization and n2c as two calls (rounds 1-4): same results
    single_variants = {}
    try:
        ref = cr.clone()
        fused = False
        one(); torch.cuda.synchronize()
        assert torch.equal(cr, ref), "single chain (two calls) differs"
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        torch.cuda.synchronize()
        single_variants["two_calls_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    except Exception as ex:
        single_variants["error"] = repr(ex)[:200]
    fused = True
    key_bytes = 8 * K * npn * L
    # ---- the same chain for B independent ciphertexts per call (cuhe_hip_mul_relin_batch): every stage runs over
    # B*np rows and a key value fetched from HBM serves four ciphertexts; results are bit-identical (checked below)
    batched = None
    try:
        B = args.relin_batch
        # B DISTINCT ciphertext pairs; every result row is compared with the single chain on the same pair
        ab = torch.randint(0, 1 << (q.logCrtPrime - 1), (B * npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        bb = torch.randint(0, 1 << (q.logCrtPrime - 1), (B * npn, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        nab = torch.empty((B * npn, L), dtype=torch.int64, device=dev); nbb = torch.empty_like(nab)
        for i in range(B):
            ck(lib.cuhe_hip_ct_ntt(nab[i * npn:].data_ptr(), ab[i * npn:].data_ptr(), logq, 0, None))
            ck(lib.cuhe_hip_ct_ntt(nbb[i * npn:].data_ptr(), bb[i * npn:].data_ptr(), logq, 0, None))
        out = torch.empty((B * npn, q.crtLen), dtype=torch.int32, device=dev)
        for _ in range(2):
            ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), nab.data_ptr(), nbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        keep_a, keep_b = na.clone(), nb.clone()
        singles = []
        for i in range(B):
            na.copy_(nab[i * npn:(i + 1) * npn]); nb.copy_(nbb[i * npn:(i + 1) * npn])
            one()
            singles.append(cr.clone())
            assert torch.equal(out[i * npn:(i + 1) * npn], cr), "batched result %d differs from the single chain" % i
        na.copy_(keep_a); nb.copy_(keep_b); one(); torch.cuda.synchronize()
        breps = max(6, 40 // B)
        t0 = time.perf_counter()
        for _ in range(breps):
            ck(lib.cuhe_hip_mul_relin_batch(out.data_ptr(), nab.data_ptr(), nbb.data_ptr(), 0, B, 0, None))
        torch.cuda.synchronize()
        bdt = (time.perf_counter() - t0) / breps / B
        # algorithmic minimum per ciphertext of a batch: the keys once per call, two ct-domain operands in, one CRT result out
        alg = key_bytes / B + 2 * 8 * npn * L + 4 * npn * q.modLen
        batched = {"value": round(1.0 / bdt, 2), "unit": "mul+relin/s", "ms_per_ciphertext": round(bdt * 1e3, 4), "batch": B,
                   "key_bytes_per_ciphertext": key_bytes // min(B, 16),
                   "algorithmic_bytes_per_ciphertext": int(alg), "frac_hbm": round(alg / bdt / 1e9 / HBM_PEAK_GBS, 4),
                   "checked": "%d distinct ciphertext pairs, every result equal to the single chain" % B,
                   "note": "B independent chains per call on one stream; products formed on load by the inverse transforms; inverse CRT column sums on the matrix cores (int8 MFMA, base-128 digits of the residue products); key-switch inner product on the matrix cores (int8 MFMA over signed base-256 digits) in tiles of 16 ciphertexts"}
        # twice the batch (the keys are amortised over more ciphertexts): the same operands twice, the two halves of the result equal
        try:
            B2 = 2 * B
            nab2, nbb2 = torch.cat((nab, nab)), torch.cat((nbb, nbb))
            out2 = torch.empty((B2 * npn, q.crtLen), dtype=torch.int32, device=dev)
            for _ in range(2):
                ck(lib.cuhe_hip_mul_relin_batch(out2.data_ptr(), nab2.data_ptr(), nbb2.data_ptr(), 0, B2, 0, None))
            torch.cuda.synchronize()
            assert torch.equal(out2[:B * npn], out) and torch.equal(out2[B * npn:], out), "batch of %d differs from the batch of %d" % (B2, B)
            t0 = time.perf_counter()
            for _ in range(max(4, breps // 2)):
                ck(lib.cuhe_hip_mul_relin_batch(out2.data_ptr(), nab2.data_ptr(), nbb2.data_ptr(), 0, B2, 0, None))
            torch.cuda.synchronize()
            b2dt = (time.perf_counter() - t0) / max(4, breps // 2) / B2
            batched["twice_the_batch"] = {"batch": B2, "ms_per_ciphertext": round(b2dt * 1e3, 4), "value": round(1.0 / b2dt, 2), "checked": "both halves equal the batch of %d" % B}
            del nab2, nbb2, out2
        except Exception as ex:
            batched["twice_the_batch"] = {"error": repr(ex)[:200]}
    except Exception as ex:
        batched = {"error": repr(ex)[:300]}
    # ---- the batched call from several host threads at once (own stream and own scratch each: the library is
    # re-entrant): the HBM-bound inner product of one call overlaps the instruction-bound transforms of another
    concurrent = None
    try:
        import threading
        T, Bc, creps = args.relin_threads, 4, 10
        bufs = []
        for _ in range(T):
            st = C.c_void_p(); ck(lib.cuhe_hip_stream_create(0, C.byref(st)))
            lo = (len(bufs) * Bc) % max(1, B - Bc + 1)                      # a different slice of the distinct pairs per thread
            bufs.append((st, nab[lo * npn:(lo + Bc) * npn].contiguous(), nbb[lo * npn:(lo + Bc) * npn].contiguous(),
                         torch.empty((Bc * npn, q.crtLen), dtype=torch.int32, device=dev), lo))
        torch.cuda.synchronize()

        def work(t, n):
            st, x, y, o, _ = bufs[t]
            for _ in range(n):
                ck(lib.cuhe_hip_mul_relin_batch(o.data_ptr(), x.data_ptr(), y.data_ptr(), 0, Bc, 0, st))
            ck(lib.cuhe_hip_stream_sync(0, st))
        for t in range(T):
            work(t, 1)
        th = [threading.Thread(target=work, args=(t, creps)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        cdt = (time.perf_counter() - t0) / (T * Bc * creps)
        for bf in bufs:
            for i in range(Bc):
                assert torch.equal(bf[3][i * npn:(i + 1) * npn], singles[bf[4] + i]), "concurrent result differs from the single chain"
        for bf in bufs:
            ck(lib.cuhe_hip_stream_destroy(0, bf[0]))
        concurrent = {"value": round(1.0 / cdt, 2), "unit": "mul+relin/s", "ms_per_ciphertext": round(cdt * 1e3, 4), "host_threads": T, "batch": Bc,
                      "note": "T host threads, one stream each, batched calls of 4 ciphertexts"}
    except Exception as ex:
        concurrent = {"error": repr(ex)[:300]}
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters(); lib.cuhe_hip_set_negacyclic(-1)
    cpu = None
    if with_cpu and not args.cyclic:
        # the same multiply + relinearise on the host cores through the oracle (reported baseline + checker of the rows above; bench_aux/cpu.py)
        try:
            from .cpu import cpu_mulrelin_baseline
            cpu = cpu_mulrelin_baseline(np, params, a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32), ek, first_rows)
        except AssertionError:
            raise
        except Exception as ex:
            cpu = {"error": repr(ex)[:300]}
    return {"value": round(1.0 / dt, 2), "unit": "mul+relin/s", "ms": round(dt * 1e3, 3), "cpu_baseline": cpu,
            "params": {"setParameters": [d, p, w, mn, cut, m], "ring_degree": q.modLen, "numCrtPrime": npn, "numEvalKey": K, "transform": rep},
            "algorithmic_bytes": key_bytes, "achieved_GBs": round(key_bytes / dt / 1e9, 1),
            "frac_hbm": round(key_bytes / dt / 1e9 / HBM_PEAK_GBS, 4), "key_upload_s": round(init_s, 2), "dispatch_last_transform": single_dispatch,
            "chain": "ct_mul ; ct_intt ; icrt ; relin_crt (relinearization + n2c as one call)", "variants": single_variants,
            "batched": batched, "concurrent": concurrent}
