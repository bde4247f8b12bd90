"""CPU tests of the CRT-prime-sharded multiply+relinearise driver (cuhe_amd/sharded.py): world_size 2 and 3
over gloo, stage functions supplied by the oracle (test infrastructure), result compared with the
unsharded oracle pipeline.  Exercises exactly the partition / padded all-gather / reassembly logic that
runs over RCCL on GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cuhe_amd.sharded import ShardedMulRelin, all_gather_rows, shard_bounds

ARGS = (3, 2, 8, 40, 20, 1155)      # toy ring: 4 primes at level 0, 3 at level 1


def test_shard_bounds_partition():
    for npr in (1, 3, 4, 7, 25, 48):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                f, c = shard_bounds(npr, world, r)
                seen += list(range(f, f + c))
                assert c in (npr // world, npr // world + 1)
            assert seen == list(range(npr))


def _t32(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int32))
def _t64(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
def _n32(t): return t.contiguous().numpy().view(np.uint32)
def _n64(t): return t.contiguous().numpy().view(np.uint64)


class OracleBackend:
    def __init__(self, o, ek):
        self.o, self.ek = o, ek

    def num_primes(self, lvl): return self.o.np_(lvl)
    def ntt_mul_rows(self, a, b): return _t64(self.o.ntt_mul(_n64(a), _n64(b)))

    def intt_mod_range(self, X, lvl, first, count):
        full = np.zeros((first + count, self.o.prm.nttLen), dtype=np.uint64)
        full[first:] = _n64(X)
        return _t32(self.o.intt_mod(full)[first:])

    def icrt(self, crt_all, lvl): return _t32(self.o.icrt(_n32(crt_all), lvl))
    def relin_range(self, raw, lvl, first, count): return _t64(self.o.relin(_n32(raw), lvl, self.ek)[first:first + count])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, lvl, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as O
        o = O.Ctx(*ARGS)
        prm = o.prm
        K, W0, M0 = prm.numEvalKey, o.words(0), o.coeff_modulus(0)
        ek_raw = np.stack([O.random_raw(prm.rawLen, prm.modLen, W0, M0, 0xE000 + j)[0] for j in range(K)])
        ek = o.init_relin(ek_raw)
        npr = o.np_(lvl)
        rng = np.random.default_rng(123)                 # same operands on every rank
        a = np.zeros((npr, prm.crtLen), dtype=np.uint32); b = np.zeros_like(a)
        for i in range(npr):
            a[i, :prm.modLen] = rng.integers(0, o.primes[i], prm.modLen)
            b[i, :prm.modLen] = rng.integers(0, o.primes[i], prm.modLen)
        na, nb = o.ntt(a), o.ntt(b)
        sh = ShardedMulRelin(OracleBackend(o, ek), lvl, rank, world)
        res_own = sh.mul_relin(_t64(na[sh.first:sh.first + sh.count]), _t64(nb[sh.first:sh.first + sh.count]))
        res_all = all_gather_rows(res_own, npr, world)
        if rank == 0:
            want = o.mul_relin_crt(a, b, lvl, ek)
            q.put(bool(np.array_equal(_n32(res_all), want)))
        o.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,lvl", [(2, 0), (3, 0), (2, 1)])
def test_sharded_mul_relin_gloo(world, lvl):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lvl, q)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True
