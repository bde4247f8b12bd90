"""one ciphertext multiply + relinearise with its CRT primes sharded over the ranks (N > 1)."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

from .record import HBM_PEAK_GBS, ROOT


def sharded_comm_guard(per_rank, world, in_library):
    """SURVEY 8(e) / cuhe/CuHE.cu:217-256: a `mul_relin_sharded` value quoted as "exchange inside the library over RCCL" is only printed
    when EVERY rank's communicator really spans the job: ncclCommCount == world, ncclCommUserRank == the rank, and the library's own
    view agrees (cuhe_hip_comm_info, gathered from all ranks).  Returns None when the record may be printed, else the reason.  (When the
    in-library communicator is not in use -- the torch.distributed exchange around the same stages -- there is nothing to check.)"""
    import re
    if not in_library:
        return None
    if len(per_rank) != world or any(not isinstance(x, str) for x in per_rank):
        return "communicator reports of %d rank(s) for a job of %d" % (sum(isinstance(x, str) for x in per_rank), world)
    for r, text in enumerate(per_rank):
        m = re.search(r"ncclCommCount (-?\d+), ncclCommUserRank (-?\d+) \(library: (-?\d+) ranks, rank (-?\d+)\)", text)
        if not m or "communicator initialised" not in text:
            return "rank %d: no initialised communicator in its report (%s)" % (r, text[:120])
        cnt, urank, lranks, lrank = (int(v) for v in m.groups())
        if cnt != world or lranks != world:
            return "rank %d: ncclCommCount %d / library %d ranks in a job of %d" % (r, cnt, lranks, world)
        if urank != r or lrank != r:
            return "rank %d reports ncclCommUserRank %d / library rank %d" % (r, urank, lrank)
    return None


def bench_mulrelin_sharded(lib, ck, torch, np, dist, dev, rank, world, args, single_dev=False):
    """primes of ONE ciphertext sharded over the ranks, one all-gather (CRT rows before ICRT) per
    multiply+relinearise; value = multiplies per second of the whole job (max time over ranks).  The whole chain, RCCL
    all-gather included, is one C-ABI call per multiply (cuhe_hip_mul_relin_sharded) enqueued on the compute stream;
    if the in-library communicator cannot be made (e.g. every rank on one device in the gloo smoke test) the exchange
    falls back to torch.distributed around the same stage functions (cuhe_amd/sharded.py)."""
    from cuhe_amd import capi
    from cuhe_amd.sharded import HipBackend, ShardedMulRelin
    d, p, w, mn, cut, m = args.relin_params
    # local set-up first; the ranks then agree (one all-reduce) that everybody is ready before the first data-path
    # collective, so that a local failure (e.g. out of memory) cannot leave the others hanging in the all-gather
    ready, err = 1, None
    try:
        lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
        ck(lib.cuhe_hip_set_parameters(d, p, w, mn, cut, m))
        ck(lib.cuhe_hip_init(None, 0))
        q = capi.get_params()
        K, W = q.numEvalKey, lib.cuhe_hip_words_coeff(0)
        ek = np.random.default_rng(7).integers(0, 1 << 32, (K, q.rawLen, W), dtype=np.uint32)
        ek[:, :, W - 1] &= 0x7FFF
        # each rank keeps the evaluation keys of ITS primes only (keys partitioned with the primes)
        kf, kc = C.c_int(), C.c_int()
        ck(lib.cuhe_hip_key_range(world, rank, C.byref(kf), C.byref(kc)))
        ck(lib.cuhe_hip_init_relin_range(ek.ctypes.data_as(C.c_void_p), kf.value, kc.value))
        lib_ct_len = lib.cuhe_hip_ct_len()
        key_bytes = kc.value * K * lib_ct_len * 8
        hb = HipBackend()
        sh = ShardedMulRelin(hb, 0, rank, world)
        gen = torch.Generator(device=dev); gen.manual_seed(5)
        a = torch.randint(0, 1 << (q.logCrtPrime - 1), (q.numCrtPrime, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        b = torch.randint(0, 1 << (q.logCrtPrime - 1), (q.numCrtPrime, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
        na = hb.ntt_rows(sh.own(a).contiguous()); nb = hb.ntt_rows(sh.own(b).contiguous())
        outc = torch.zeros((sh.count, q.crtLen), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
    except Exception as ex:
        ready, err = 0, repr(ex)[:200]
    flag = torch.tensor([ready], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
        return {"error": "set-up failed on at least one rank" + (": " + err if err else "")}
    # in-library communicator: rank 0 makes the id, torch.distributed carries the 128 bytes
    # (byte 128 carries rank 0's verdict: if it could not make the id NOBODY calls comm_init -- the others would block in it)
    idt = torch.zeros(129, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid = (C.c_uint8 * 128)()
        # every rank on ONE device (the gloo smoke test): RCCL refuses duplicate devices, and a communicator set-up that one rank
        # has left while the other still waits in it has no time-out -- it is not attempted at all
        made = (not single_dev) and lib.cuhe_hip_comm_unique_id(uid) == 0
        idt = torch.tensor(list(uid) + [1 if made else 0], dtype=torch.uint8, device=dev)
    dist.broadcast(idt, 0)
    host_id = [int(v) for v in idt.cpu().tolist()]
    ok = host_id[128]
    if ok:
        uid = (C.c_uint8 * 128)(*host_id[:128])
        ok = 1 if lib.cuhe_hip_comm_init(world, rank, uid) == 0 else 0
    comm_err = None if ok else ("every rank on one device" if single_dev else lib.cuhe_hip_last_error().decode()[:200])
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    in_library = int(flag.item()) == 1
    # what RCCL itself says on every rank (version, ncclCommCount, ncclCommUserRank) next to the library's view: gathered to
    # every rank so that rank 0 can print it -- the first thing to read if the first N > 1 contact misbehaves
    cbuf = C.create_string_buffer(512)
    lib.cuhe_hip_comm_info(cbuf, 512)
    my_info = "rank %d: comm_init %s; %s" % (rank, "ok" if ok else "FAILED (%s)" % comm_err, cbuf.value.decode())
    per_rank = [None] * world
    try:
        dist.all_gather_object(per_rank, my_info)
    except Exception as ex:
        per_rank = [my_info, "all_gather_object failed: %r" % (ex,)]
    if not in_library:
        lib.cuhe_hip_comm_destroy()
        if comm_err is None:
            comm_err = "comm_init failed on another rank"

    def one():
        if in_library:
            ck(lib.cuhe_hip_mul_relin_sharded(outc.data_ptr(), na.data_ptr(), nb.data_ptr(), 0, 0, None))
            return outc
        return sh.mul_relin(na, nb)
    if in_library:
        # the first call through RCCL is allowed to fail (the exact failing call and RCCL's message arrive in the error
        # string): every rank then falls back to the torch.distributed exchange together, and the leg still delivers
        lib_err = None
        try:
            first = one().clone()
            torch.cuda.synchronize()
        except Exception as ex:
            lib_err = repr(ex)[:300]
        flag = torch.tensor([0 if lib_err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            in_library = False
            comm_err = "first cuhe_hip_mul_relin_sharded failed: " + (lib_err or "on another rank")
            lib.cuhe_hip_comm_destroy()
    if in_library:                                     # same rows through the torch.distributed exchange: must agree
        assert torch.equal(first, sh.mul_relin(na, nb)), "in-library all-gather differs from the torch.distributed path"
        lib.cuhe_hip_comm_info(cbuf, 512)
        per_rank[rank] = per_rank[rank] + " | after the first call: " + cbuf.value.decode()
    else:
        first = one().clone()
    for _ in range(3):
        one()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # the part every participant repeats (ICRT of the gathered rows, window extraction, the k window transforms): timed as
    # ICRT + the key switch onto ONE prime; its share of the call is the serial fraction that bounds the speed-up
    rows_all = torch.randint(0, 1 << (q.logCrtPrime - 1), (q.numCrtPrime, q.crtLen), dtype=torch.int32, device=dev, generator=gen)
    raw = torch.zeros((q.rawLen, W), dtype=torch.int32, device=dev)
    acc1 = torch.empty((1, lib.cuhe_hip_ct_len()), dtype=torch.int64, device=dev)
    def replicated():
        ck(lib.cuhe_hip_icrt(raw.data_ptr(), rows_all.data_ptr(), lib.cuhe_hip_log_coeff(0), 0, None))
        ck(lib.cuhe_hip_relin_range(acc1.data_ptr(), raw.data_ptr(), 0, sh.first, 1, 0, None))
    for _ in range(3):
        replicated()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        replicated()
    torch.cuda.synchronize()
    t_rep = (time.perf_counter() - t0) / reps
    lib_comm_size = lib.cuhe_hip_comm_size() if in_library else None
    lib.cuhe_hip_comm_destroy()
    lib.cuhe_hip_shutdown(); lib.cuhe_hip_reset_parameters()
    refused = sharded_comm_guard(per_rank, world, in_library)
    if refused:                                         # no number under a claim that did not hold
        return {"value": None, "error": "not reported: " + refused, "rccl_per_rank": per_rank, "comm_size": lib_comm_size}
    return {"value": round(1.0 / dt, 2), "unit": "mul+relin/s (one ciphertext, primes sharded)", "ms": round(dt * 1e3, 3),
            "key_bytes_per_rank": key_bytes, "key_primes_per_rank": kc.value,
            "replicated_ms": round(t_rep * 1e3, 3), "serial_fraction": round(t_rep / dt, 3),
            "serial_note": "ICRT + window extraction + the k window transforms are repeated on every rank (exchanging the transformed windows instead "
                           "would move k*n*8 = %d B per multiply against %d B of CRT rows); the key-switch inner product, both inverse transforms and the "
                           "key memory divide by the number of ranks" % (K * lib_ct_len * 8, q.numCrtPrime * q.crtLen * 4),
            "primes_per_rank": sh.count, "numCrtPrime": q.numCrtPrime, "numEvalKey": K, "ring_degree": q.modLen,
            "comm_size": lib_comm_size, "rccl_per_rank": per_rank,
            "exchange": "RCCL all-gather inside cuhe_hip_mul_relin_sharded, on the compute stream (one in-place ncclAllGather when the blocks are equal, one padded ncclAllGather otherwise; the path every rank took is in rccl_per_rank)" if in_library
                        else "torch.distributed all-gather around the C-ABI stages (in-library communicator unavailable: %s)" % comm_err,
            "collective": "1 all-gather of %d B per rank per multiply" % (sh.count * q.crtLen * 4)}
