"""CRT-prime-sharded ciphertext multiply + relinearise.

The reference's multi-GPU mode is task parallelism (whole ciphertexts per GPU,
`cuhe/CuHE.cu:217-256`); sharding one ciphertext across GPUs is new here.  One
process per GPU; rank r owns a contiguous block of the level's CRT primes.  Every
per-prime stage (pointwise product, INTT + reduction mod Phi_m, the key-switch
inner product, the final INTT) runs on the owner with no communication; the ONLY
exchange is one all-gather of the CRT residue rows before ICRT (each coefficient
needs all of its residues).  ICRT and the window transforms are recomputed on
every rank (cheaper than a second exchange at these sizes).

The stage functions come from a backend object so that the partition / gather
logic can be exercised on CPU with gloo (tests/test_sharded_gloo.py, oracle
backend) and runs on GPUs with RCCL through the C ABI (`HipBackend`).
"""
import torch
import torch.distributed as dist


def shard_bounds(num_primes, world, rank):
    """contiguous, balanced: the first (num_primes % world) ranks own one extra prime"""
    base, extra = divmod(num_primes, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def all_gather_rows(own_rows, num_primes, world, group=None):
    """own_rows: [count_r, width] tensor of this rank's rows -> [num_primes, width] on every rank.
    Ranks may own different row counts: rows are padded to the largest shard for the collective."""
    if world == 1:
        return own_rows
    width = own_rows.shape[1]
    if num_primes % world == 0:
        # equal shards: one flat collective straight into the result, no padding and no reassembly copies
        out = torch.empty((num_primes, width), dtype=own_rows.dtype, device=own_rows.device)
        try:
            dist.all_gather_into_tensor(out, own_rows.contiguous(), group=group)
            return out
        except (RuntimeError, NotImplementedError, AttributeError):
            pass                                    # backend without the flat form: the general path below
    maxc = (num_primes + world - 1) // world
    pad = torch.zeros((maxc, width), dtype=own_rows.dtype, device=own_rows.device)
    pad[: own_rows.shape[0]] = own_rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty((num_primes, width), dtype=own_rows.dtype, device=own_rows.device)
    for r in range(world):
        f, c = shard_bounds(num_primes, world, r)
        out[f:f + c] = parts[r][:c]
    return out


class ShardedMulRelin:
    """cAnd + relin (cuhe/CuHE.cu:101,570-581) with the level's primes split over `world` ranks."""

    def __init__(self, backend, lvl, rank, world, group=None):
        self.b, self.lvl, self.rank, self.world, self.group = backend, lvl, rank, world, group
        self.num_primes = backend.num_primes(lvl)
        self.first, self.count = shard_bounds(self.num_primes, world, rank)

    def own(self, full_rows):
        return full_rows[self.first:self.first + self.count]

    def mul_relin(self, na_own, nb_own):
        """NTT-domain operand rows of the owned primes -> reduced CRT-domain result rows of the owned primes."""
        b, lvl, f, c = self.b, self.lvl, self.first, self.count
        prod = b.ntt_mul_rows(na_own, nb_own)
        crt_own = b.intt_mod_range(prod, lvl, f, c)                       # x2r: n2c (isProd)
        crt_all = all_gather_rows(crt_own, self.num_primes, self.world, self.group)   # the one exchange
        raw = b.icrt(crt_all, lvl)                                        #      c2r  (replicated)
        acc = b.relin_range(raw, lvl, f, c)                               # windows replicated, keys of owned primes only
        return b.intt_mod_range(acc, lvl, f, c)                           # n2c (isProd)


class HipBackend:
    """stage functions on the GPU through the C ABI; tensors are torch int32/int64 views of u32/u64."""

    def __init__(self, dev_index=0):
        from . import capi
        self.capi, self.lib, self.ck = capi, capi.lib, capi.check
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.prm = capi.get_params()
        self.ctlen = self.lib.cuhe_hip_ct_len()          # row length of NTT-domain (ct) rows: nttLen, or modLen on x^n + 1 rings

    def num_primes(self, lvl):
        return self.lib.cuhe_hip_num_crt_prime(lvl)

    def ntt_mul_rows(self, a, b):
        z = torch.empty_like(a)
        self.ck(self.lib.cuhe_hip_ntt_mul_rows(z.data_ptr(), a.data_ptr(), b.data_ptr(), a.shape[0], 0, None))
        return z

    def intt_mod_range(self, X, lvl, first, count):
        out = torch.zeros((count, self.prm.crtLen), dtype=torch.int32, device=self.dev)
        self.ck(self.lib.cuhe_hip_intt_mod_range(out.data_ptr(), X.data_ptr(), lvl, first, count, 0, None))
        return out

    def icrt(self, crt_all, lvl):
        raw = torch.zeros((self.prm.rawLen, self.lib.cuhe_hip_words_coeff(lvl)), dtype=torch.int32, device=self.dev)
        self.ck(self.lib.cuhe_hip_icrt(raw.data_ptr(), crt_all.data_ptr(), self.lib.cuhe_hip_log_coeff(lvl), 0, None))
        return raw

    def relin_range(self, raw, lvl, first, count):
        out = torch.empty((count, self.ctlen), dtype=torch.int64, device=self.dev)
        self.ck(self.lib.cuhe_hip_relin_range(out.data_ptr(), raw.data_ptr(), lvl, first, count, 0, None))
        return out

    def ntt_rows(self, crt_rows):
        out = torch.empty((crt_rows.shape[0], self.ctlen), dtype=torch.int64, device=self.dev)
        self.ck(self.lib.cuhe_hip_ntt_rows(out.data_ptr(), crt_rows.data_ptr(), crt_rows.shape[0], 0, None))
        return out
