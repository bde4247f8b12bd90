"""CPU: a semantic pin of the oracle's CRT / transform / relinearisation / modulus-switch restatement.  The reference holds
no vectors for these stages individually; what it does hold is the END-TO-END meaning -- its DHS example
must decrypt correctly (examples/DHS/simple_DHS.cu checkAnd :130).  This test runs that meaning through the oracle
alone: a DHS/LTV key pair, evaluation keys and ciphertexts are built here with Python integers (nothing from oracle/),
the homomorphic AND goes through the oracle's stages (crt, ntt, pointwise product, INTT + reduction, ICRT, windowed
relinearisation, modulus switch), and the result must decrypt -- again with Python integers -- to the product of the
messages.  A wrong window decomposition, key layout, Barrett reduction or modswitch rounding makes decryption fail."""
import random

import numpy as np

import oracle_lib as O


def _cyclotomic(m):
    def mu(n):
        r, p = 1, 2
        while p * p <= n:
            if n % p == 0:
                n //= p
                if n % p == 0:
                    return 0
                r = -r
            p += 1
        return -r if n > 1 else r
    a = [1]
    for d in range(1, m + 1):
        if m % d == 0 and mu(m // d) == 1:                       # multiply by x^d - 1
            b = [0] * (len(a) + d)
            for i, c in enumerate(a):
                b[i + d] += c
                b[i] -= c
            a = b
    for d in range(1, m + 1):
        if m % d == 0 and mu(m // d) == -1:                      # exact division by x^d - 1
            q = [0] * (len(a) - d)
            r = a[:]
            for k in range(len(a) - 1, d - 1, -1):
                q[k - d] = r[k]
                r[k - d] += r[k]
                r[k] = 0
            a = q
    return a


def _polymul_mod(a, b, phi, q):
    n = len(phi) - 1
    t = [0] * (2 * n - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(2 * n - 2, n - 1, -1):
        c = t[k]
        if c:
            for i in range(n + 1):
                t[k - n + i] -= c * phi[i]
    return [v % q for v in t[:n]]


def _inverse_mod_prime(f, phi, p):
    """f^-1 in F_p[x]/(phi) by the extended Euclidean algorithm, or None"""
    def trim(v):
        while v and v[-1] == 0:
            v.pop()
        return v
    r0, r1 = trim([c % p for c in phi]), trim([c % p for c in f])
    t0, t1 = [], [1]
    while r1:
        lead = pow(r1[-1], p - 2, p)
        while len(r0) >= len(r1):
            sh, c = len(r0) - len(r1), r0[-1] * lead % p
            for i, v in enumerate(r1):
                r0[i + sh] = (r0[i + sh] - c * v) % p
            if len(t0) < len(t1) + sh:
                t0 += [0] * (len(t1) + sh - len(t0))
            for i, v in enumerate(t1):
                t0[i + sh] = (t0[i + sh] - c * v) % p
            trim(r0)
            if not r0:
                break
        r0, r1, t0, t1 = r1, r0, t1, t0
    if len(r0) != 1:
        return None
    g = pow(r0[0], p - 2, p)
    n = len(phi) - 1
    t = t0 + [0] * max(0, n - len(t0))
    for k in range(len(t) - 1, n - 1, -1):
        c = t[k]
        if c:
            for i in range(n + 1):
                t[k - n + i] = (t[k - n + i] - c * phi[i]) % p
    return [v * g % p for v in t[:n]]


class Scheme:
    """a DHS / LTV key pair, evaluation keys, encryption and decryption with Python integers only (nothing from oracle/
    except the parameter set and the CRT primes the context reports): examples/DHS/DHS.cu keygen / encrypt / decrypt"""

    def __init__(self, o, seed):
        q = o.prm
        self.o, self.n, self.K, self.w = o, q.modLen, q.numEvalKey, q.logRelin
        n = self.n
        self.phi = _cyclotomic(q.mSize)
        assert len(self.phi) == n + 1 and self.phi[-1] == 1
        self.qs = [o.coeff_modulus(l) for l in range(q.depth)]
        q0 = self.qs[0]
        primes = [int(p) for p in o.primes]
        prod = 1
        for p in primes: prod *= p
        assert prod == q0 and len(primes) == q.numCrtPrime
        self.rnd = rnd = random.Random(seed)
        tern = self.tern = lambda: [rnd.choice((-1, 0, 1)) for _ in range(n)]
        while True:                                             # f = 2 f' + 1 invertible modulo every prime
            f = [2 * v for v in tern()]; f[0] += 1
            invs = [_inverse_mod_prime(f, self.phi, p) for p in primes]
            if all(v is not None for v in invs):
                break
        self.f = f
        finv = [0] * n
        for p, row in zip(primes, invs):
            mi = q0 // p
            lift = mi * pow(mi % p, p - 2, p)
            for k in range(n):
                finv[k] = (finv[k] + lift * row[k]) % q0
        assert _polymul_mod([v % q0 for v in f], finv, self.phi, q0) == [1] + [0] * (n - 1)
        self.pk = [2 * v % q0 for v in _polymul_mod([v % q0 for v in tern()], finv, self.phi, q0)]
        ek = [[(a + (fk << (self.w * j))) % q0 for a, fk in zip(self.enc([0] * n, 0), f)] for j in range(self.K)]
        self.ek_raw = np.stack([O.ints_to_raw(e, q.rawLen, o.words(0)) for e in ek])
        self.keys = o.init_relin(self.ek_raw)

    def enc(self, msg, lvl):
        """coefficients of the ciphertext of the message polynomial `msg` at level lvl"""
        mod = self.qs[lvl]
        hs = _polymul_mod([v % mod for v in self.pk], [v % mod for v in self.tern()], self.phi, mod)
        return [(a + 2 * e + b) % mod for a, e, b in zip(hs, self.tern(), msg)]

    def enc_crt(self, msg, lvl):
        q = self.o.prm
        return self.o.crt(O.ints_to_raw(self.enc(msg, lvl), q.rawLen, self.o.words(lvl)), lvl)

    def dec_crt(self, crt_rows, lvl):
        """(message polynomial mod 2, largest |noise| coefficient) of CRT rows of a level-lvl ciphertext that went through
        `mults` multiplications: c * f^(degree) -- here always degree 1 after relinearisation"""
        mod = self.qs[lvl]
        c = O.raw_to_ints(self.o.icrt(crt_rows, lvl), self.n)
        dec = _polymul_mod(c, [v % mod for v in self.f], self.phi, mod)
        cen = [v - mod if v > (mod - 1) // 2 else v for v in dec]
        return [v % 2 for v in cen], max(abs(v) for v in cen)


def test_oracle_stages_carry_a_homomorphic_and_to_the_right_plaintext():
    d, pm, w, mn, cut, m = 3, 2, 8, 40, 20, 1155
    o = O.Ctx(d, pm, w, mn, cut, m)
    try:
        S = Scheme(o, 20260926)
        n, q1 = S.n, S.qs[1]
        m1 = [S.rnd.randrange(2) for _ in range(n)]; m2 = [S.rnd.randrange(2) for _ in range(n)]
        a, b = S.enc_crt(m1, 0), S.enc_crt(m2, 0)
        prod = o.mul_relin_crt(a, b, 0, S.keys)                  # cAnd ; relin   (CRT rows, level 0)
        low = o.modswitch(prod)                                  # modSwitch      (level 1)
        got, noise = S.dec_crt(low, 1)
        want = [v % 2 for v in _polymul_mod(m1, m2, S.phi, 1 << 40)]
        assert got == want
        # and the noise is where the scheme says it is: far below q1 / 2
        assert noise < q1 >> 12
    finally:
        o.close()
