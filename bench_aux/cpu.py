"""CPU baselines of the bench record: the oracle (`kind: "port"`) timed on the GPU box's host cores, bounded samples.  The oracle is test
infrastructure: it is imported here as the reported baseline and as the checker of the GPU results, never on the product path."""
import os
import sys
import time

from .record import ROOT


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def cpu_ntt_baseline(np, L, sample=0, seconds=4.0):
    """64K-point forward transforms per second on all host cores through the oracle's throughput form (orc_ntt_ext_fast_batch: the radix-2
    butterflies of the oracle with the reduction the reference's field arithmetic uses -- the fold 2^64 = 2^32 - 1, no division --, tables made
    once per length, OpenMP over transforms): "the same algorithm on the host cores".  The division form the record carried until round 5
    (oracle transform with u128 % P, tables per transform) is timed beside it on a smaller sample."""
    O = _oracle()
    cores = O.host_cores()                                        # affinity mask capped by the cgroup CPU quota, not os.cpu_count()
    group = max(cores * 8, 32)                                   # transforms per call (outputs: group x L x 8 bytes)
    xh = np.random.default_rng(1).integers(0, 1 << 32, (group, L // 2), dtype=np.uint32)
    got, used = O.ntt_ext_fast_batch(xh, L, 0)                    # warm-up: tables, thread pool, page faults
    assert np.array_equal(got[0], O.ntt_ext(xh[0], L)) and np.array_equal(got[-1], O.ntt_ext(xh[-1], L)), "throughput form differs from the oracle transform"
    done, t0 = 0, time.perf_counter()
    while True:
        O.ntt_ext_fast_batch(xh, L, 0)
        done += group
        dt = time.perf_counter() - t0
        if (sample and done >= sample) or (not sample and dt >= seconds):
            break
    rec = {"value": round(done / dt, 1), "unit": "NTT/s", "cores": used, "logical_cpus_visible": os.cpu_count(), "kind": "port",
           "sample": "%d 64K-point forward transforms in %.1f s: oracle radix-2 transform with the Solinas-fold reduction (no division), tables per length, OpenMP over transforms"
                     % (done, dt)}
    small = xh[:max(cores, 8)]
    O.ntt_ext_batch(small[:2], L, 0)
    t1 = time.perf_counter()
    _, used2 = O.ntt_ext_batch(small, L, 0)
    d2 = time.perf_counter() - t1
    rec["oracle_transform_per_call_tables"] = {"value": round(len(small) / d2, 1), "unit": "NTT/s", "cores": used2,
                                               "sample": "%d transforms, tables rebuilt per transform (the form the parity tests call), %.1f s" % (len(small), d2)}
    return rec


def cpu_mulrelin_baseline(np, params, a, b, ek_raw, gpu_rows, reps=3):
    """ciphertext multiply + relinearise of ONE pair on the host cores: the oracle's chain per prime through the negacyclic restatement
    (orc_nc_mul_relin_prepared: keys transformed beforehand like the GPU's resident keys, OpenMP over primes / windows / coefficients),
    and the check that the GPU's single chain produced exactly these rows.  a, b: u32[np][crtLen] reduced CRT rows; ek_raw: the raw keys
    handed to cuhe_hip_init_relin; gpu_rows: what the timed GPU chain returned for (a, b)."""
    O = _oracle()
    o = O.Ctx(*params)
    used = O.set_threads(0)
    try:
        t0 = time.perf_counter()
        ekc = o.key_residues(ek_raw)
        h = o.nc_prepare(0, ekc)
        prep = time.perf_counter() - t0
        try:
            want = o.nc_mul_relin_prepared(h, a, b, 0)            # warm-up + the checker
            q = o.prm
            same = bool(np.array_equal(want[:, :q.modLen], gpu_rows[:, :q.modLen]))
            t1 = time.perf_counter()
            for _ in range(reps):
                o.nc_mul_relin_prepared(h, a, b, 0)
            dt = (time.perf_counter() - t1) / reps
        finally:
            o.nc_prepared_free(h)
    finally:
        O.set_threads(1)
        o.close()
    assert same, "GPU multiply + relinearise differs from the oracle chain"
    return {"value": round(1.0 / dt, 3), "unit": "mul+relin/s", "cores": used, "kind": "port", "gpu_rows_equal_oracle": same,
            "sample": "%d multiplies + relinearisations of one ciphertext pair (oracle chain per prime on x^n + 1, keys transformed beforehand in %.1f s, OpenMP), %.2f s each"
                      % (reps, prep, dt)}
