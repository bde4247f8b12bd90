"""CPU: the first round of the homomorphic PRINCE evaluation carried through the ORACLE's stages alone, checked against
the round state the reference's own example holds (examples/Prince/Prince.cu:108-145, first entry; fixture
tests/golden/prince_kat.json).  This pins the oracle's relinearisation (window decomposition, key layout), modulus
switching and ICRT to a reference-held value without the HIP path in between: the GPU tests check the same known answer
through the HIP kernels, and HIP == oracle stage by stage.

Keys, encryption and decryption are Python integers (tests/test_oracle_dhs_semantics.py: Scheme); the circuit -- whitening
XOR, the 16 S-boxes from the algebraic normal form of the S-box table (6 + 4 multiplications per S-box over two
levels, each followed by relinearisation and a modulus switch, as examples/Prince/Prince.cu:204-322 schedules them) -- runs
on oracle.c's crt_add / mul_relin_crt / modswitch.  The plaintext state does not depend on the ring, so the small ring of
the toy parameter set (depth 3) is used: the whole test takes seconds."""
import json
import os

import numpy as np

import oracle_lib as O
from test_oracle_dhs_semantics import Scheme

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SBOX = [0xB, 0xF, 0x3, 0x2, 0xA, 0xC, 0x9, 0x1, 0x6, 0x7, 0x8, 0x0, 0xE, 0x5, 0xD, 0x4]      # PRINCE S-box (Borghoff et al. 2012)


def anf(box):
    """anf[o][mask] = coefficient of the monomial `mask` (8 = a ... 1 = d, a the most significant input bit) of output bit o"""
    out = []
    for o in range(4):
        t = [(box[x] >> (3 - o)) & 1 for x in range(16)]
        for i in range(4):                                    # Moebius transform
            for x in range(16):
                if x & (1 << i): t[x] ^= t[x ^ (1 << i)]
        out.append(t)
    return out


def test_anf_reproduces_the_sbox():
    A = anf(SBOX)
    for x in range(16):
        y = 0
        for o in range(4):
            bit = 0
            for mask in range(16):
                if A[o][mask] and (x & mask) == mask: bit ^= 1
            y = (y << 1) | bit
        assert y == SBOX[x]


def test_prince_round_one_through_oracle_stages():
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "prince_kat.json")))
    pt, k0, k1 = int(kat["plaintext"], 16), int(kat["k0"], 16), int(kat["k1"], 16)
    want = kat["round_states_bits"][0]
    o = O.Ctx(3, 2, 8, 40, 20, 1155)
    try:
        S = Scheme(o, 0x9fb5)
        n = S.n
        const = lambda b: [b] + [0] * (n - 1)
        bits = lambda v: [(v >> (63 - i)) & 1 for i in range(64)]
        A = anf(SBOX)
        # level-0 ciphertexts of the plaintext bits and of the whitening-key bits; RC_0 = 0 and k1 enter as constants
        state = [o.crt_add(S.enc_crt(const(p), 0), S.enc_crt(const(k), 0)) for p, k in zip(bits(pt), bits(k0))]
        state = [o.crt_add_int(c, b) for c, b in zip(state, bits(k1))]
        AND = lambda x, y, lvl: o.modswitch(o.mul_relin_crt(x, y, lvl, S.keys))          # cAnd ; relin ; modSwitch
        out_bits, worst = [], 0
        for nib in range(16):
            v = state[4 * nib: 4 * nib + 4]                                              # a, b, c, d at level 0
            idx = {8: 0, 4: 1, 2: 2, 1: 3}
            deg2 = {}
            for i, mi in enumerate((8, 4, 2, 1)):
                for mj in (8, 4, 2, 1)[i + 1:]:
                    deg2[mi | mj] = AND(v[idx[mi]], v[idx[mj]], 0)                       # level 1
            v1 = [o.modswitch(x) for x in v]                                             # the inputs at level 1
            deg3 = {}
            for mask in (14, 13, 11, 7):
                hi = mask & -mask                                                        # lowest variable times the pair of the others
                deg3[mask] = AND(deg2[mask ^ hi], v1[idx[hi]], 1)                        # level 2
            deg2_2 = {k: o.modswitch(x) for k, x in deg2.items()}
            v2 = [o.modswitch(x) for x in v1]
            for ob in range(4):
                acc = None
                for mask in range(1, 16):
                    if not A[ob][mask]: continue
                    term = v2[idx[mask]] if mask in idx else deg2_2[mask] if mask in deg2_2 else deg3.get(mask)
                    assert term is not None, "degree-4 monomial in the PRINCE S-box?"
                    acc = term if acc is None else o.crt_add(acc, term)
                if A[ob][0]: acc = o.crt_add_int(acc, 1)
                msg, noise = S.dec_crt(acc, 2)
                assert not any(msg[1:]), "the plaintext is a constant polynomial"
                out_bits.append(msg[0]); worst = max(worst, noise)
        assert "".join(str(b) for b in out_bits) == want
        assert worst < S.qs[2] >> 4                                                      # decryption was not a coincidence of wrapped noise
    finally:
        o.close()
