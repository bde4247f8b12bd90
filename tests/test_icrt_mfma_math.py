"""CPU model of the matrix-core inverse CRT (cuhe_amd/csrc/icrt_mfma.cuh): the digit tables exactly as
cuhe_context.hip lays them out, the operand / result lane map of v_mfma_i32_32x32x32_i8, the word assembly with signed
carries, the join of the two lane halves and the conditional subtraction through NM = 2^(32 NW) - M -- in Python integers,
against  sum_i t_i (M / p_i) mod M.  It pins the ARITHMETIC of the kernel (bounds included); the kernel itself is compared
with the VALU kernel and the oracle on the GPU (tests/test_gpu_parity.py)."""
import random

import pytest


def _is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2; s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _primes_below(bits, count):
    out, n = [], (1 << bits) - 1
    while len(out) < count:
        if _is_prime(n):
            out.append(n)
        n -= 2
    return out


class Model:
    def __init__(self, primes):
        self.P = primes
        self.np = len(primes)
        self.M = 1
        for p in primes:
            self.M *= p
        self.W = (self.M.bit_length() + 31) // 32
        self.tiles, self.ks = (self.W + 1 + 7) // 8, (self.np + 7) // 8
        self.NW, self.WH = 8 * self.tiles, 4 * self.tiles
        ND = 4 * self.NW
        self.mi = [self.M // p for p in primes]
        self.b = [pow(self.mi[i] % primes[i], -1, primes[i]) for i in range(self.np)]
        self.bq = [(self.b[i] << 32) // primes[i] for i in range(self.np)]
        self.dg = {}
        for i in range(self.np):                                 # signed base-256 digits of 128^a (M / p_i)
            for a in range(4):
                v, carry, o = self.mi[i] << (7 * a), 0, []
                for d in range(ND):
                    byte = ((v >> (8 * d)) & 0xFF) + carry
                    carry = 1 if byte >= 128 else 0
                    o.append(byte - 256 if carry else byte)
                assert carry == 0 and sum(o[d] << (8 * d) for d in range(ND)) == v
                self.dg[(i, a)] = o
        nm = (1 << (32 * self.NW)) - self.M
        self.nm = [(nm >> (32 * k)) & 0xFFFFFFFF for k in range(self.NW)]

    def first_operand(self, m, s, lane):
        """16 bytes of lane `lane` for result tile m, K step s: [prime 8 s + 4 (lane >> 5) + e][digit a]"""
        rho, hk = lane & 31, lane >> 5
        d = 4 * (((rho >> 2) & 1) * self.WH + 4 * m + (rho >> 3)) + (rho & 3)
        return [[self.dg[(8 * s + 4 * hk + e, a)][d] if 8 * s + 4 * hk + e < self.np else 0 for a in range(4)] for e in range(4)]

    def residue_product(self, x, i):
        """Shoup form of x b mod p for any x < 2^32 (the kernel's three multiplies and a min)"""
        p = self.P[i]
        r = (x * self.b[i] - ((x * self.bq[i]) >> 32) * p) & 0xFFFFFFFF
        assert r < 2 * p
        return min(r, (r - p) & 0xFFFFFFFF)

    def icrt(self, x):
        t = [self.residue_product(x[i], i) for i in range(self.np)]
        assert t == [x[i] * self.b[i] % self.P[i] for i in range(self.np)]
        alpha = 0.0
        for i in range(self.np):
            alpha += t[i] / self.P[i]
        q = int(max(alpha - 2.0 ** -30, 0.0))
        acc = {}
        for m in range(self.tiles):                              # D[row][col] = sum over K of A[row][k] B[k][col], one column here
            for rho in range(32):
                tot = 0
                for s in range(self.ks):
                    for hk in range(2):
                        A = self.first_operand(m, s, rho + 32 * hk)
                        for e in range(4):
                            i = 8 * s + 4 * hk + e
                            tv = t[i] if i < self.np else 0
                            for a in range(4):
                                tot += A[e][a] * ((tv >> (7 * a)) & 0x7F)
                assert abs(tot) < 1 << 23
                acc[(m, rho)] = tot
        wd, carries = [[0] * self.WH for _ in range(2)], [0, 0]
        for h in range(2):                                       # lane (c, h): register r of tile m is row (r & 3) + 8 (r >> 2) + 4 h
            carry = 0
            for m in range(self.tiles):
                for r2 in range(4):
                    jj = 4 * m + r2
                    o = [acc[(m, r1 + 8 * r2 + 4 * h)] for r1 in range(4)]
                    plo, phi = o[0] + (o[1] << 8), o[2] + (o[3] << 8)
                    assert -2 ** 31 <= plo < 2 ** 31 and -2 ** 31 <= phi < 2 ** 31
                    col = q * self.nm[h * self.WH + jj] + plo + (phi << 16) + carry
                    assert -2 ** 63 <= col < 2 ** 63
                    wd[h][jj], carry = col & 0xFFFFFFFF, col >> 32
            carries[h] = carry
        cc = carries[0]
        for jj in range(self.WH):
            v = wd[1][jj] + cc
            wd[1][jj], cc = v & 0xFFFFFFFF, v >> 32
        d, cy = [[0] * self.WH for _ in range(2)], [0, 0]
        for h in range(2):
            c = 0
            for jj in range(self.WH):
                v = wd[h][jj] + self.nm[h * self.WH + jj] + c
                d[h][jj], c = v & 0xFFFFFFFF, v >> 32
            cy[h] = c
        cc = cy[0]
        for jj in range(self.WH):
            v = d[1][jj] + cc
            d[1][jj], cc = v & 0xFFFFFFFF, v >> 32
        out = d if (cy[1] | cc) else wd
        return sum(out[h][jj] << (32 * (h * self.WH + jj)) for h in range(2) for jj in range(self.WH))


@pytest.mark.parametrize("np_,bits", [(48, 24), (5, 25), (13, 28), (33, 23)])
def test_matrix_core_icrt_model(np_, bits):
    mdl = Model(_primes_below(bits, np_))
    M, P = mdl.M, mdl.P
    rng = random.Random(np_)
    cases = [0, 1, 2, M - 1, M - 2, M // 2, M // np_, M // np_ + 1, 3 * (M // np_) - 1] + [rng.randrange(M) for _ in range(3)]
    for v in cases:
        assert mdl.icrt([v % p for p in P]) == v
    x = [rng.randrange(1 << 32) for _ in P]                       # unreduced residues
    x[0] = 0xFFFFFFFF
    assert mdl.icrt(x) == sum((x[i] * mdl.b[i] % P[i]) * mdl.mi[i] for i in range(np_)) % M
