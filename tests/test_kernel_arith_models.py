"""CPU models of three small pieces of kernel arithmetic introduced in round 4, in Python integers: they pin the bounds the
kernels rely on (the GPU tests compare the kernels themselves with the oracle).
  * k_modswitch (ops_kernels.cuh): the multiple of p that makes `src - dirty` non-negative comes from a shift;
  * k_crt<ACC64>: the sums  sum_k word_k (2^(32k) mod p)  fit 64 bits exactly when the launcher's test says so;
  * mod_small: floor(2^64 / p) as reciprocal gives x mod p for the operands these kernels feed it."""
import random

import pytest


def clz32(x):
    return 32 - x.bit_length()


def mod_small(x, p):
    """the device function: q = floor(x m / 2^64), m = floor(2^64 / p); one conditional subtraction"""
    m = (1 << 64) // p
    r = x - ((x * m) >> 64) * p
    assert 0 <= r < 2 * p
    return r - p if r >= p else r


PRIMES = [1048573, 16777213, 16777183, 33554393, 8388593, 524287, 2147483647, 65537, 268435399]


@pytest.mark.parametrize("p", PRIMES)
def test_modswitch_lift_and_reductions(p):
    lift = p << (11 + clz32(p))
    assert lift % p == 0 and (1 << 42) <= lift < (1 << 43)
    rng = random.Random(p)
    for modmsg in (2, 3, 16, 1024):
        for _ in range(200):
            ptl = rng.choice(PRIMES)
            src, d = rng.randrange(1 << 32), rng.randrange(ptl)
            ep = d % modmsg
            dirty = d
            if ep:
                dirty += -ep * ptl if d > (ptl - 1) // 2 else ep * ptl
            x = src - dirty + lift
            assert 0 <= x < (1 << 64)
            t = mod_small(x, p)
            assert t == (src - dirty) % p
            inv = rng.randrange(p)
            assert mod_small(t * inv, p) == t * inv % p


@pytest.mark.parametrize("W,p", [(24, 16777213), (36, 16777213), (35, 8388593), (104, 33554393), (4, 2147483647), (128, 33554393)])
def test_crt_sums_fit_64_bits_when_the_launcher_says_so(W, p):
    acc64 = W * p <= (1 << 32)                       # launch_crt's test (cuhe_keyswitch.hip)
    worst = sum(0xFFFFFFFF * (p - 1) for _ in range(W))
    assert (worst < (1 << 64)) or not acc64
    if acc64:
        rng = random.Random(W)
        words = [rng.randrange(1 << 32) for _ in range(W)]
        c = [pow(2, 32 * k, p) for k in range(W)]
        s = sum(w * ck for w, ck in zip(words, c))
        assert s < (1 << 64) and mod_small(s, p) == sum(w << (32 * k) for k, w in enumerate(words)) % p
