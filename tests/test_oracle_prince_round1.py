"""CPU: the first TWO rounds of the homomorphic PRINCE evaluation carried through the ORACLE's stages alone, checked against
the round states the reference's own example holds (examples/Prince/Prince.cu:108-145, first two entries; fixture
tests/golden/prince_kat.json).  This pins the oracle's relinearisation (window decomposition, key layout), modulus
switching and ICRT to a reference-held value without the HIP path in between: the GPU tests check the same known answer
through the HIP kernels, and HIP == oracle stage by stage.

Keys, encryption and decryption are Python integers (tests/test_oracle_dhs_semantics.py: Scheme); the circuit -- whitening
XOR, the 16 S-boxes from the algebraic normal form of the S-box table (6 + 4 multiplications per S-box over two
levels, each followed by relinearisation and a modulus switch, as examples/Prince/Prince.cu:204-322 schedules them) -- runs
on oracle.c's crt_add / mul_relin_crt / modswitch.  The plaintext state does not depend on the ring, so a small ring (the
toy cyclotomic of depth 5) is used.  After round state 0 the evaluation goes one round further: linear layer (M', shift
rows: XORs of ciphertexts), round constant, second S-box layer, down to the reference's round state 1 -- sixteen S-boxes
with DIFFERENT inputs (round state 0 is 0100 sixteen times: every S-box of the first layer sees the input 0xF).  ~45 s."""
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


RC1 = 0x13198a2e03707344                                # round constant 1 (PRINCE specification)
SR = [0, 5, 10, 15, 4, 9, 14, 3, 8, 13, 2, 7, 12, 1, 6, 11]


def m_prime_sources():
    """three source bits per output bit of M': block-diagonal (M^0, M^1, M^1, M^0), 16x16 blocks of 4x4 identities with one
    diagonal entry cleared (PRINCE specification, section 3)"""
    src = [[] for _ in range(64)]
    first = (0, 1, 1, 0)
    for chunk in range(4):
        for br in range(4):
            for r in range(4):
                for bc in range(4):
                    if r != (first[chunk] + br + bc) % 4:
                        src[16 * chunk + 4 * br + r].append(16 * chunk + 4 * bc + r)
    return src


def sbox_layer(o, S, A, state, lvl):
    """the 16 S-boxes on ciphertexts of level lvl -> ciphertexts of level lvl + 2 (Prince.cu:204-322: 6 + 4 multiplications
    per S-box over two levels, each followed by relinearisation and a modulus switch)"""
    AND = lambda x, y, l: o.modswitch(o.mul_relin_crt(x, y, l, S.keys))
    idx = {8: 0, 4: 1, 2: 2, 1: 3}
    out = []
    for nib in range(16):
        v = state[4 * nib: 4 * nib + 4]
        deg2 = {}
        for i, mi in enumerate((8, 4, 2, 1)):
            for mj in (8, 4, 2, 1)[i + 1:]:
                deg2[mi | mj] = AND(v[idx[mi]], v[idx[mj]], lvl)
        v1 = [o.modswitch(x) for x in v]
        deg3 = {}
        for mask in (14, 13, 11, 7):
            hi = mask & -mask
            deg3[mask] = AND(deg2[mask ^ hi], v1[idx[hi]], lvl + 1)
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
            out.append(acc)
    return out


def test_prince_round_two_through_oracle_stages():
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "prince_kat.json")))
    pt, k0, k1 = int(kat["plaintext"], 16), int(kat["k0"], 16), int(kat["k1"], 16)
    o = O.Ctx(5, 2, 8, 40, 20, 1155)
    try:
        S = Scheme(o, 0x35fc)
        n = S.n
        const = lambda b: [b] + [0] * (n - 1)
        bits = lambda v: [(v >> (63 - i)) & 1 for i in range(64)]
        A = anf(SBOX)
        state = [o.crt_add(S.enc_crt(const(p), 0), S.enc_crt(const(k), 0)) for p, k in zip(bits(pt), bits(k0))]
        state = [o.crt_add_int(c, b) for c, b in zip(state, bits(k1))]
        state = sbox_layer(o, S, A, state, 0)                                            # level 2: round state 0
        got0 = "".join(str(S.dec_crt(c, 2)[0][0]) for c in state)
        assert got0 == kat["round_states_bits"][0]
        src = m_prime_sources()
        mixed = []
        for i in range(64):                                                              # M': three ciphertext XORs per bit
            acc = state[src[i][0]]
            for j in src[i][1:]: acc = o.crt_add(acc, state[j])
            mixed.append(acc)
        shifted = [mixed[4 * SR[i // 4] + i % 4] for i in range(64)]                     # shift rows: a permutation of the nibbles
        state = [o.crt_add_int(c, b) if b else c for c, b in zip(shifted, bits(RC1 ^ k1))]
        nibbles = set()
        for i in range(16):
            nibbles.add(tuple(S.dec_crt(state[4 * i + k], 2)[0][0] for k in range(4)))
        assert len(nibbles) >= 8                                                         # the second layer's S-boxes see many different inputs
        state = sbox_layer(o, S, A, state, 2)                                            # level 4: round state 1
        out_bits, worst = [], 0
        for c in state:
            msg, noise = S.dec_crt(c, 4)
            assert not any(msg[1:]), "the plaintext is a constant polynomial"
            out_bits.append(msg[0]); worst = max(worst, noise)
        want = kat["round_states_bits"][1]
        assert "".join(str(b) for b in out_bits) == want
        assert len({want[4 * i: 4 * i + 4] for i in range(16)}) >= 8                    # (the reference-held state itself: many different nibbles)
        assert worst < S.qs[4] >> 4
    finally:
        o.close()
