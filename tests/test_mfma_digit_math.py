"""CPU model of the arithmetic behind the matrix-core key-switch inner product (cuhe_amd/csrc/ops_kernels.cuh:
signed_digits, k_relin_mac_mfma, fold_digits), in Python integers: the signed base-256 digits of a
representative of each 64-bit factor, the int32 accumulators per digit diagonal, and the fold of the 15 diagonal sums back
into Z_P with 2^64 = 2^32 - 1 and 2^96 = -1.  Pins the algorithm (ranges, bounds, identities) where there is no GPU;
the kernel itself is compared bit for bit with the vector-unit kernel and the oracle in tests/test_gpu_parity.py."""
import random

P = (1 << 64) - (1 << 32) + 1
C = 0x8080808080808080
T = 0x7F7F7F7F7F7F7F80
M64 = (1 << 64) - 1


def signed_digits(a):
    """the kernel's formula on a u64: bytes of ((a + C + (a >= T ? 2^32 - 1 : 0)) mod 2^64) xor C, as signed bytes"""
    y = (a + C + (0xFFFFFFFF if a >= T else 0)) & M64
    w = y ^ C
    return [((w >> (8 * l)) & 0xFF) - 256 * (((w >> (8 * l)) & 0xFF) >> 7) for l in range(8)]


BIAS = 1 << 24
BIAS_MOD_P = sum(BIAS << (8 * t) for t in range(15)) % P


def fold_digits(D):
    """sum_t D[t] 256^t mod P as the kernel does it: the accumulators start at 2^24, four words of positive 25-bit numbers,
    one unsigned 128-bit sum with the constant 2P - (bias mod P), then lo + hi (2^32 - 1)"""
    E = [d + BIAS for d in D]
    assert all(0 < x < 1 << 25 for x in E)
    w = [sum(E[4 * q + r] << (8 * r) for r in range(4) if 4 * q + r < 15) for q in range(4)]
    assert all(x < 1 << 50 for x in w)
    U = (2 * P - BIAS_MOD_P) + w[0] - w[2] - w[3] + ((w[1] + w[2]) << 32)
    assert 0 <= U < 1 << 84
    lo, hi = U & M64, U >> 64
    assert hi < 1 << 20
    r = lo + hi * 0xFFFFFFFF                                           # mad_eps: one correction makes it canonical
    assert r < 2 * P
    return r - P if r >= P else r


def test_signed_digits_represent_the_value_mod_p():
    rng = random.Random(1)
    edge = [0, 1, T - 1, T, T + 1, P - 1, P, P + 1, M64, 1 << 63, (1 << 63) - 1, C, C - 1]
    for a in edge + [rng.getrandbits(64) for _ in range(20000)]:
        d = signed_digits(a)
        assert all(-128 <= x <= 127 for x in d)
        v = sum(x << (8 * l) for l, x in enumerate(d))
        assert v % P == a % P and -C <= v < (1 << 64) - C, hex(a)


def test_digit_products_by_diagonal_fold_to_the_inner_product():
    rng = random.Random(2)
    for k in (1, 5, 40, 72, 128):
        for _ in range(60):
            worst = rng.random() < 0.2
            a = [rng.choice([T - 1, M64, 0x7F7F7F7F7F7F7F7F, C + 0x7F7F7F7F7F7F7F7F & M64]) if worst else rng.getrandbits(64) for _ in range(k)]
            e = [rng.choice([T - 1, P - 1, 0x7F7F7F7F7F7F7F7F]) if worst else rng.randrange(P) for _ in range(k)]
            D = [0] * 15
            for x, y in zip(a, e):
                dx, dy = signed_digits(x), signed_digits(y)
                for la in range(8):
                    for lb in range(8):
                        D[la + lb] += dx[la] * dy[lb]
            assert all(abs(t) < 1 << 24 for t in D), k                 # |D_t| < 8 k 2^14 = 2^24 at k = 128 (a digit -128 meets at most 127s): the biased sums stay in (0, 2^25)
            want = sum(x * y for x, y in zip(a, e)) % P
            assert fold_digits(D) == want, k
