"""CPU: the dataflow model of the one-workgroup transform (tests/onewg_model.py: thread mapping, LDS addressing, the two
half-exchanges, the folded twiddles of the zero-padded form) equals the oracle's transform (oracle.c: orc_ntt_ext,
restating cuhe/Base.cu:309-437).  The HIP kernel mirrors the model's index formulas; this test pins the formulas."""
import numpy as np
import pytest

import onewg_model as M
import oracle_lib as O


@pytest.mark.parametrize("R", [4, 8, 16])
def test_half_mode_equals_zero_padded_transform(R):
    Lh = 32 * 32 * R
    x = O.splitmix_u32_below(Lh, (1 << 32) - 1, 7 + R)
    want = O.ntt_ext(x, 2 * Lh)                      # the reference contract: u32[L/2] -> u64[L]
    u = [int(v) for v in x]
    for h in (0, 1):
        got = M.simulate(R, u, half=True, h=h)
        assert all(int(want[2 * k + h]) == got[k] for k in range(Lh)), "half %d" % h


def test_full_mode_equals_cyclic_transform_of_full_input():
    # a full-length input of Lh points: compare with the even outputs of the zero-padded transform of 2 Lh points
    R = 8
    Lh = 32 * 32 * R
    x = O.splitmix_u32_below(Lh, (1 << 32) - 1, 99)
    want = O.ntt_ext(x, 2 * Lh)
    got = M.simulate(R, [int(v) for v in x])
    assert all(int(want[2 * k]) == got[k] for k in range(Lh))


def test_32k_balanced_exchanges_equal_zero_padded_transform():
    Lh = 32768
    x = O.splitmix_u32_below(Lh, (1 << 32) - 1, 5)
    want = O.ntt_ext(x, 2 * Lh)
    u = [int(v) for v in x]
    for h in (0, 1):
        got = M.simulate32(u, half=True, h=h)
        assert all(int(want[2 * k + h]) == got[k] for k in range(Lh)), "half %d" % h
    assert (M.LDS_WORDS32 + 1024) * 8 <= 160 * 1024


def test_16k_balanced_exchanges_equal_zero_padded_transform():
    """the persistent form of the 32K-point rows: two workgroups of this geometry per CU"""
    Lh = 16384
    x = O.splitmix_u32_below(Lh, (1 << 32) - 1, 6)
    want = O.ntt_ext(x, 2 * Lh)
    u = [int(v) for v in x]
    for h in (0, 1):
        got = M.simulate_balanced(16, u, half=True, h=h)
        assert all(int(want[2 * k + h]) == got[k] for k in range(Lh)), "half %d" % h
    assert 2 * ((M.lds_words_bal(16) + 512) * 8 + 16) <= 160 * 1024
    assert M.lds_words_bal(16) * 8 >= 4 * Lh                  # the exchange buffer takes the u32 samples of the next half (LDS-DMA)


def test_lds_budget():
    # bytes of the exchange buffer + the stage-2 twiddle table: 4 / 2 / 1 workgroups per CU inside 160 KiB
    for R, per_cu in ((4, 7), (8, 4), (16, 2)):
        bytes_ = (M.lds_words(R) + 32 * R) * 8
        assert bytes_ * per_cu <= 160 * 1024, (R, bytes_)
