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


def test_oracle_stages_carry_a_homomorphic_and_to_the_right_plaintext():
    d, pm, w, mn, cut, m = 3, 2, 8, 40, 20, 1155
    o = O.Ctx(d, pm, w, mn, cut, m)
    try:
        q = o.prm
        n, K = q.modLen, q.numEvalKey
        phi = _cyclotomic(m)
        assert len(phi) == n + 1 and phi[-1] == 1
        q0, q1 = o.coeff_modulus(0), o.coeff_modulus(1)
        # the CRT primes of level 0: q0 is their product, level 1 drops the last one
        primes, rest, last = [], q0, q0 // q1
        cand = 1 << 21
        while rest > 1:
            cand -= 1
            if rest % cand == 0:
                primes.append(cand); rest //= cand
        assert last in primes and len(primes) == q.numCrtPrime
        rnd = random.Random(20260926)
        tern = lambda: [rnd.choice((-1, 0, 1)) for _ in range(n)]
        while True:                                             # f = 2 f' + 1 invertible modulo every prime
            f = [2 * v for v in tern()]; f[0] += 1
            invs = [_inverse_mod_prime(f, phi, p) for p in primes]
            if all(v is not None for v in invs):
                break
        finv = [0] * n
        for p, row in zip(primes, invs):
            mi = q0 // p
            lift = mi * pow(mi % p, p - 2, p)
            for k in range(n):
                finv[k] = (finv[k] + lift * row[k]) % q0
        assert _polymul_mod([v % q0 for v in f], finv, phi, q0) == [1] + [0] * (n - 1)
        pk = [2 * v % q0 for v in _polymul_mod([v % q0 for v in tern()], finv, phi, q0)]
        enc = lambda msg, mod: [(a + 2 * e + b) % mod for a, e, b in zip(_polymul_mod(pk, [v % mod for v in tern()], phi, mod), tern(), msg)]
        ek = [[(a + (fk << (w * j))) % q0 for a, fk in zip(enc([0] * n, q0), f)] for j in range(K)]
        ek_raw = np.stack([O.ints_to_raw(e, q.rawLen, o.words(0)) for e in ek])
        keys = o.init_relin(ek_raw)
        m1 = [rnd.randrange(2) for _ in range(n)]; m2 = [rnd.randrange(2) for _ in range(n)]
        c1, c2 = enc(m1, q0), enc(m2, q0)
        a = o.crt(O.ints_to_raw(c1, q.rawLen, o.words(0)), 0)
        b = o.crt(O.ints_to_raw(c2, q.rawLen, o.words(0)), 0)
        prod = o.mul_relin_crt(a, b, 0, keys)                    # cAnd ; relin   (CRT rows, level 0)
        low = o.modswitch(prod)                                  # modSwitch      (level 1)
        c = O.raw_to_ints(o.icrt(low, 1), n)
        dec = _polymul_mod(c, [v % q1 for v in f], phi, q1)
        got = [(v - q1 if v > (q1 - 1) // 2 else v) % 2 for v in dec]
        want = [v % 2 for v in _polymul_mod(m1, m2, phi, 1 << 40)]
        assert got == want
        # and the noise is where the scheme says it is: far below q1 / 2
        assert max(abs(v - q1 if v > q1 // 2 else v) for v in dec) < q1 >> 12
    finally:
        o.close()
