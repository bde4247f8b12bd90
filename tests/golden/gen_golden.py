#!/usr/bin/env python3
"""Golden-vector generator for the cuHE hot path -- pure Python big integers.

Independent of oracle/ and of cuhe_amd/: it restates the *mathematical meaning*
of each reference stage with Python ints (and sympy.isprime for primality) so
that the C oracle can be pinned against something that shares no code with it.
The reference has no Python and cannot be built here (CUDA + NTL), so nothing
is imported from /root/reference; every expected value below follows a formula
the reference documents:

  * field / transform:  tests/test_ntt.cu:38-64 (by-definition DFT, root g),
                        cuhe/Base.cu:489,656,841 (L^-1 constants)
  * parameters:         cuhe/Parameters.cu:53-145
  * primes / moduli:    cuhe/Operations.cu:37-134
  * mulZZX meaning:     examples/DHS/DHS.cu:219-221  (a*b % Phi_m, coeffs % q)
  * modswitch:          cuhe/Base.cu:1112-1138
  * relinearisation:    cuhe/Relinearization.cu:76-88, cuhe/Base.cu:361-371

Run:  python tests/golden/gen_golden.py   (writes tests/golden/*.json)
"""
import hashlib
import json
import os
import sys

from sympy import isprime, totient

HERE = os.path.dirname(os.path.abspath(__file__))
P = 0xFFFFFFFF00000001
G = 15893793146607301539
MASK64 = (1 << 64) - 1


# ---------------------------------------------------------------- generator
class SplitMix:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)


def u32_below(n, bound, seed):
    g = SplitMix(seed)
    return [g.next() % bound for _ in range(n)]


def random_big(modlen, words, q, seed):
    """matches tests/oracle_lib.random_raw"""
    nw = words + 1
    draws = u32_below(modlen * nw, 0xFFFFFFFF, seed)
    out = []
    for i in range(modlen):
        v = 0
        for k in range(nw):
            v |= draws[i * nw + k] << (32 * k)
        out.append(v % q)
    return out


def sha_u32(vals):
    h = hashlib.sha256()
    h.update(b"".join(int(v).to_bytes(4, "little") for v in vals))
    return h.hexdigest()


def sha_u64(vals):
    h = hashlib.sha256()
    h.update(b"".join(int(v).to_bytes(8, "little") for v in vals))
    return h.hexdigest()


def sha_big(vals, words):
    h = hashlib.sha256()
    h.update(b"".join(int(v).to_bytes(4 * words, "little") for v in vals))
    return h.hexdigest()


# ---------------------------------------------------------------- transform
def fft(a, w):
    n = len(a)
    if n == 1:
        return a
    e = fft(a[0::2], w * w % P)
    o = fft(a[1::2], w * w % P)
    out = [0] * n
    t = 1
    h = n // 2
    for k in range(h):
        v = o[k] * t % P
        out[k] = (e[k] + v) % P
        out[k + h] = (e[k] - v) % P
        t = t * w % P
    return out


def ntt_ext(x, L):
    w = pow(G, 65536 // L, P)
    return fft(list(x[:L // 2]) + [0] * (L // 2), w)


def ntt_by_definition(x, L, idxs):
    w = pow(G, 65536 // L, P)
    r = [1] * L
    for i in range(1, L):
        r[i] = r[i - 1] * w % P
    out = {}
    for i in idxs:
        acc = 0
        for j in range(L // 2):
            acc += x[j] * r[(i * j) % L]
        out[i] = acc % P
    return out


def intt_modp(X, L, p):
    w = pow(G, 65536 // L, P)
    winv = pow(w, P - 2, P)
    y = fft(list(X), winv)
    li = pow(L, P - 2, P)
    return [(v * li % P) % p for v in y]


# ---------------------------------------------------------------- parameters
def numbits(x):
    return int(x).bit_length()


def isqrt(n):
    import math
    return math.isqrt(n)


def set_param(d, p, w, mn, cut, m):
    q = dict(depth=d, modMsg=p, logRelin=w, logCoeffMin=mn, logCoeffCut=cut, mSize=m)
    q["logCoeffMax"] = mn + cut * (d - 1)
    q["modLen"] = int(totient(m)) if m >= 3 else m
    q["modLen2"] = max(8192, 1 << numbits(q["modLen"] - 1))
    q["rawLen"] = q["crtLen"] = q["modLen2"]
    q["nttLen"] = 2 * q["modLen2"]
    q["logMsg"] = numbits(p - 1)
    q["wordsMsg"] = (q["logMsg"] + 31) // 32
    q["numEvalKey"] = (q["logCoeffMax"] + w - 1) // w if w else 0
    lcp = numbits(isqrt(P // q["modLen"]))
    ncp = (mn + lcp - 1) // lcp
    lcp = 0
    while lcp * ncp < mn:
        lcp += 1
    q["logCrtPrime"] = lcp
    q["numCrtPrime"] = ncp + d - 1
    return q


def gen_crt_primes(q):
    pnum, d, l = q["numCrtPrime"], q["depth"], q["logCrtPrime"]
    logmid = q["logCoeffMin"] - (pnum - d) * l
    primes = [0] * pnum
    temp = (1 << l) - 1
    for i in range(0, pnum - d):
        while not isprime(temp):
            temp -= 1
        primes[i] = temp
        temp -= 1
    tmid = (1 << logmid) - 1 if logmid != l else temp
    while not isprime(tmid):
        tmid -= 1
    primes[pnum - d] = tmid
    if q["logCoeffCut"] == logmid:
        temp = tmid - 1
    elif q["logCoeffCut"] == l:
        temp -= 1
    else:
        temp = (1 << q["logCoeffCut"]) - 1
    for i in range(pnum - d + 1, pnum):
        while (not isprime(temp)) or temp % q["modMsg"] != 1:
            temp -= 1
        primes[i] = temp
        temp -= 1
    return primes


def log_coeff(q, lvl):
    if lvl == -1:
        return q["logMsg"]
    if lvl < q["depth"]:
        return q["logCoeffMax"] - lvl * q["logCoeffCut"]
    return q["logCoeffMin"] - q["logCrtPrime"]


def words_coeff(q, lvl):
    return max(1, (log_coeff(q, lvl) + 31) // 32)


def num_eval_key(q, lvl):
    return (log_coeff(q, lvl) + q["logRelin"] - 1) // q["logRelin"]


# ---------------------------------------------------------------- polynomials
def cyclotomic(m):
    from sympy import Poly, cyclotomic_poly
    from sympy.abc import x
    c = Poly(cyclotomic_poly(m, x), x).all_coeffs()
    return [int(v) for v in reversed(c)]


def kron_mul(a, b, bits):
    """product of non-negative integer polynomials via Kronecker substitution"""
    A = 0
    for i in reversed(range(len(a))):
        A = (A << bits) | a[i]
    B = 0
    for i in reversed(range(len(b))):
        B = (B << bits) | b[i]
    C_ = A * B
    mask = (1 << bits) - 1
    out = []
    for _ in range(len(a) + len(b) - 1):
        out.append(C_ & mask)
        C_ >>= bits
    return out


def poly_rem(f, mod):
    """remainder of integer polynomial f modulo monic mod (low-to-high lists)"""
    f = list(f)
    n = len(mod) - 1
    nz = [(i, c) for i, c in enumerate(mod[:-1]) if c]
    for k in range(len(f) - 1, n - 1, -1):
        c = f[k]
        if c:
            f[k] = 0
            for i, mc in nz:
                f[k - n + i] -= c * mc
    return f[:n] + [0] * max(0, n - len(f))


def poly_rem_cyclo(f, m, phi):
    """same, using Phi_m | x^m - 1 to fold first (keeps the long division short)"""
    f = list(f)
    if len(f) > m:
        for k in range(len(f) - 1, m - 1, -1):
            f[k - m] += f[k]
        f = f[:m]
    return poly_rem(f, phi)


def mul_mod(a, b, m, phi, q, coeff_bits):
    n = len(phi) - 1
    prod = kron_mul(a[:n], b[:n], 2 * coeff_bits + n.bit_length() + 2)
    r = poly_rem_cyclo(prod, m, phi)
    return [v % q for v in r]


# ---------------------------------------------------------------- stage semantics
def modswitch(res, primes, invp, modmsg):
    """res: list over primes of residue lists; drops the last prime."""
    npn = len(primes)
    pt = primes[-1]
    n = len(res[0])
    out = [[0] * n for _ in range(npn - 1)]
    for idx in range(n):
        dirty = res[npn - 1][idx]
        ep = dirty % modmsg
        if ep != 0:
            if dirty > (pt - 1) // 2:
                dirty -= ep * pt
            else:
                dirty += ep * pt
        for i in range(npn - 1):
            out[i][idx] = ((res[i][idx] - dirty) % primes[i]) * invp[(npn - 1, i)] % primes[i]
    return out


def window(v, w, j, W):
    """cuhe/Base.cu:361-371: only words wi, wi+1 are read"""
    wi = (w * j) >> 5
    words = [(v >> (32 * t)) & 0xFFFFFFFF for t in range(W)]
    s = words[wi] | ((words[wi + 1] << 32) if wi + 1 < W else 0)
    s >>= (w * j) & 31
    return s & ((1 << w) - 1)


# ---------------------------------------------------------------- fixtures
def fx_field():
    g = SplitMix(0xF1E1D)
    cases = []
    for _ in range(64):
        x, y = g.next(), g.next()
        cases.append(dict(x=str(x), y=str(y), add=str((x + y) % P), sub=str((x - y) % P),
                          mul=str((x * y) % P)))
    edge = [0, 1, P - 1, P, P + 1, MASK64, 0xFFFFFFFF, 1 << 32, (1 << 32) - 1]
    for x in edge:
        for y in edge:
            cases.append(dict(x=str(x), y=str(y), add=str((x + y) % P), sub=str((x - y) % P),
                              mul=str((x * y) % P)))
    shifts = []
    for a in range(8):
        for b in range(8):
            x = g.next()
            shifts.append(dict(x=str(x), l=3 * a * b, out=str((x << (3 * a * b)) % P)))
    consts = dict(
        g_pow_65536=str(pow(G, 65536, P)), g_pow_32768=str(pow(G, 32768, P)), g_pow_1024=str(pow(G, 1024, P)),
        inv_16384=str(pow(16384, P - 2, P)), inv_32768=str(pow(32768, P - 2, P)), inv_65536=str(pow(65536, P - 2, P)))
    return dict(P=str(P), g=str(G), cases=cases, shifts=shifts, consts=consts)


def fx_ntt():
    out = {}
    for L in (16384, 32768, 65536):
        seed = 0xC0FFEE + L
        x = u32_below(L // 2, 1 << 31, seed)           # test_ntt.cu:117 uses rand() (31-bit)
        X = ntt_ext(x, L)
        idxs = [0, 1, 2, 3, 5, 64, 511, 1023, 1024, 4097, L // 2 - 1, L // 2, L // 2 + 1, L - 2, L - 1]
        bydef = ntt_by_definition(x, L, idxs)
        for i in idxs:
            assert bydef[i] == X[i], "fast transform disagrees with the definition"
        p = 2097143
        back = intt_modp(X, L, p)
        assert back[:L // 2] == [v % p for v in x] and not any(back[L // 2:])
        # full 32-bit inputs as well (largest admissible word)
        x2 = u32_below(L // 2, 0xFFFFFFFF, seed + 1)
        X2 = ntt_ext(x2, L)
        out[str(L)] = dict(seed=seed, bound=1 << 31, by_definition={str(i): str(bydef[i]) for i in idxs},
                           sha256_full=sha_u64(X), head=[str(v) for v in X[:8]],
                           seed32=seed + 1, sha256_full32=sha_u64(X2),
                           intt_prime=p, intt_sha256=sha_u32(back))
    return out


PARAM_SETS = {
    "dhs_simple": (5, 2, 1, 61, 20, 8191),       # examples/DHS/simple_DHS.cu:218
    "prince": (25, 2, 16, 25, 25, 21845),         # examples/Prince/Prince.cu:48-49
    "toy1155": (3, 2, 8, 40, 20, 1155),           # composite m, tiny ring inside a 16K transform
    "pow2_16384": (3, 2, 16, 50, 25, 16384),      # Phi = x^8192 + 1
    "c3_65536": (9, 2, 16, 576, 24, 65536),       # BASELINE config 3 shape: n = 2^15, L = 65536, 32 primes < 2^24
    "c4_65536": (25, 2, 16, 576, 24, 65536),      # BASELINE config 4 shape: 64K-point transforms, 48 primes < 2^24
    # BASELINE config 1 exactly: N = 2^13 with ONE CRT prime (p = 2097143), on x^8192 + 1 and on Phi_8191
    "c1_pow2_1prime": (1, 2, 16, 21, 21, 16384),
    "c1_prime_m_1prime": (1, 2, 16, 21, 21, 8191),
}


def fx_params():
    out = {}
    for name, args in PARAM_SETS.items():
        q = set_param(*args)
        primes = gen_crt_primes(q)
        d, pnum = q["depth"], q["numCrtPrime"]
        moduli = []
        for lvl in range(d):
            M = 1
            for j in range(pnum - lvl):
                M *= primes[j]
            moduli.append(hex(M))
        invp = []
        for i in range(1, pnum):
            for j in range(i):
                invp.append(pow(primes[i] % primes[j], -1, primes[j]))
        lv = {}
        for lvl in [-1] + list(range(d + 1)):
            lv[str(lvl)] = dict(logCoeff=log_coeff(q, lvl), wordsCoeff=words_coeff(q, lvl))
        for lvl in range(d):
            lv[str(lvl)]["numEvalKey"] = num_eval_key(q, lvl) if q["logRelin"] else 0
            lv[str(lvl)]["numCrtPrime"] = pnum - lvl
        out[name] = dict(args=list(args), params=q, primes=primes, coeff_moduli=moduli,
                         invp_sha256=sha_u32(invp), invp_head=invp[:6], levels=lv)
    return out


def fx_pipeline(name, lvl_list, with_relin, full_vectors):
    args = PARAM_SETS[name]
    q = set_param(*args)
    primes = gen_crt_primes(q)
    m, n = q["mSize"], q["modLen"]
    phi = cyclotomic(m)
    assert len(phi) == n + 1
    pnum = q["numCrtPrime"]
    invp = {}
    for i in range(1, pnum):
        for j in range(i):
            invp[(i, j)] = pow(primes[i] % primes[j], -1, primes[j])
    out = dict(args=list(args), levels={})
    for lvl in lv_iter(lvl_list):
        npn = pnum - lvl
        M = 1
        for j in range(npn):
            M *= primes[j]
        W = words_coeff(q, lvl)
        sa, sb = 0xA000 + 17 * lvl, 0xB000 + 31 * lvl
        a = random_big(n, W, M, sa)
        b = random_big(n, W, M, sb)
        rec = dict(seed_a=sa, seed_b=sb, words=W, num_primes=npn)
        # CRT meaning: residues
        crt_a = [[v % p for v in a] for p in primes[:npn]]
        rec["crt_a_sha256"] = sha_u32([v for row in crt_a for v in row + [0] * (q["crtLen"] - n)])
        # mulZZX meaning
        c = mul_mod(a, b, m, phi, M, 32 * W)
        rec["mul_sha256"] = sha_big(c + [0] * (q["rawLen"] - n), W)
        if full_vectors:
            rec["mul_hex"] = [hex(v) for v in c]
        # CRT-domain add / not / add-plain
        cs = [[(x + y) % p for x, y in zip(ra, rb)] for ra, rb, p in
              zip(crt_a, [[v % p for v in b] for p in primes[:npn]], primes[:npn])]
        rec["crt_add_sha256"] = sha_u32([v for row in cs for v in row + [0] * (q["crtLen"] - n)])
        # modswitch on CRT(a)
        if npn >= 2 and lvl < q["depth"] - 1:
            ms = modswitch(crt_a, primes[:npn], invp, q["modMsg"])
            rec["modswitch_sha256"] = sha_u32([v for row in ms for v in row + [0] * (q["crtLen"] - n)])
            # semantic check for modMsg = 2: result == (a - delta)/p_t mod q' with delta = a mod p_t, parity fixed
            if q["modMsg"] == 2:
                pt = primes[npn - 1]
                for idx in (0, 1, n - 1):
                    d_ = a[idx] % pt
                    if d_ % 2:
                        d_ = d_ - pt if d_ > (pt - 1) // 2 else d_ + pt
                    t = (a[idx] - d_) // pt
                    assert (a[idx] - d_) % pt == 0
                    for i in range(npn - 1):
                        assert ms[i][idx] == t % primes[i]
            if full_vectors:
                rec["modswitch_head"] = [row[:4] for row in ms]
        out["levels"][str(lvl)] = rec
    if with_relin:
        lvl = 0
        npn = pnum
        W0 = words_coeff(q, 0)
        M0 = 1
        for j in range(pnum):
            M0 *= primes[j]
        K = q["numEvalKey"]
        w = q["logRelin"]
        eks = [random_big(n, W0, M0, 0xE000 + j) for j in range(K)]
        relin = {}
        for lv in lv_iter(lvl_list):
            npn = pnum - lv
            Mq = 1
            for j in range(npn):
                Mq *= primes[j]
            Wl = words_coeff(q, lv)
            k = num_eval_key(q, lv)
            ct = random_big(n, Wl, Mq, 0xC100 + lv)
            acc = [0] * (2 * n - 1)
            for j in range(k):
                wj = [window(v, w, j, Wl) for v in ct]
                pr = kron_mul(wj, eks[j], 32 * W0 + w + n.bit_length() + 2)
                for t, v in enumerate(pr):
                    acc[t] += v
            red = poly_rem_cyclo(acc, m, phi)
            rows = [[v % p for v in red] for p in primes[:npn]]
            # window recomposition sanity: sum_j window_j * 2^(w j) == ct (low bits)
            for idx in (0, n - 1):
                rec_v = sum(window(ct[idx], w, j, Wl) << (w * j) for j in range(k))
                assert rec_v == ct[idx] % (1 << (w * k)) or w * k > 32 * Wl
            relin[str(lv)] = dict(seed_ct=0xC100 + lv, num_keys=k,
                                  crt_sha256=sha_u32([v for row in rows for v in row + [0] * (q["crtLen"] - n)]),
                                  head=[row[:4] for row in rows] if full_vectors else None)
        out["relin"] = dict(seed_ek_base=0xE000, levels=relin)
    return out


def lv_iter(l):
    return list(l)


def main():
    sets = {}
    if "--config1" in sys.argv:                      # only the fixtures added for BASELINE config 1 (+ the parameter table)
        sets["params.json"] = fx_params()
        for nm in ("c1_pow2_1prime", "c1_prime_m_1prime"):
            print("pipeline", nm, "...", flush=True)
            sets["pipeline_%s.json" % nm] = fx_pipeline(nm, [0], False, False)
        for fn, obj in sets.items():
            with open(os.path.join(HERE, fn), "w") as f:
                json.dump(obj, f, indent=1)
            print("wrote", fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")
        return
    print("field ...", flush=True)
    sets["field.json"] = fx_field()
    print("ntt ...", flush=True)
    sets["ntt.json"] = fx_ntt()
    print("params ...", flush=True)
    sets["params.json"] = fx_params()
    print("pipeline toy1155 ...", flush=True)
    sets["pipeline_toy1155.json"] = fx_pipeline("toy1155", [0, 1, 2], True, True)
    print("pipeline pow2_16384 ...", flush=True)
    sets["pipeline_pow2_16384.json"] = fx_pipeline("pow2_16384", [0, 2], False, False)
    print("pipeline dhs_simple ...", flush=True)
    sets["pipeline_dhs_simple.json"] = fx_pipeline("dhs_simple", [0, 4], False, False)
    for nm in ("c1_pow2_1prime", "c1_prime_m_1prime"):
        print("pipeline", nm, "...", flush=True)
        sets["pipeline_%s.json" % nm] = fx_pipeline(nm, [0], False, False)
    for fn, obj in sets.items():
        with open(os.path.join(HERE, fn), "w") as f:
            json.dump(obj, f, indent=1)
        print("wrote", fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    sys.setrecursionlimit(100000)
    main()
