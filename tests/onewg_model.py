"""Thread-level model of the one-workgroup transform (cuhe_amd/csrc/ntt_onewg.cuh): which thread holds which value in
each register stage, where it writes it in LDS, who reads it back.  Test infrastructure only (tests/test_onewg_model.py
checks it against the oracle); the HIP kernel mirrors these index formulas line by line.

A sub-transform of Lh = 32 * R * 32 points (R = 8 / 16 / 32: 8K / 16K / 32K) is done by ONE workgroup of T = 32 R threads,
32 values per thread, in three register stages separated by two exchanges through LDS:

  stage 1  thread m (< T):            x[a] = u[a T + m], a < 32;  A[ka] = DFT32_a(x) * w_Lh^(m ka)
  X1       -> thread t2 = c + 32 kq:  y[i][b] = A_{m = 32 b + c}[ka],  ka = kq + R i,  i < 32 / R,  b < R
  stage 2  B[i][kb] = DFT_R_b(y[i]) * w_T^(c kb)
  X2       -> thread t3 = ka + 32 kb: z[c] = B_{(kq, c), i}[kb]
  stage 3  Y[t3 + T kc] = DFT32_c(z)

Each exchange moves HALF of every thread's 32 values at a time (the LDS holds half a transform):
  X1, half h:  the writers' A[ka], ka in [16 h, 16 h + 16)  ->  buf[((ka - 16 h) * 32 + c) * (R + 1) + b]
  X2, half h:  the writers' B[.][kb], kb in [R/2 h, R/2 (h+1))  ->  buf[((kb - R/2 h) * 32 + ka) * 33 + c]

R = 32 (32K points, ONE workgroup per CU) uses exchanges whose READ side is unconditional and balanced (simulate32):
  stage-2 thread t2 = kq + 32 c (kq in the lane bits);
  X1, half h:  writers with b in [16 h, 16 h + 16) (whole waves: t = 32 b + c) store all 32 A[ka]
               ->  buf[c * 545 + ka * 17 + (b - 16 h)],  every reader (kq, c) takes its 16 values b
  X2, half h:  writers with c in [16 h, 16 h + 16) (whole waves: t2 = kq + 32 c) store all 32 B[kb]
               ->  buf[kb * 544 + ka * 17 + (c - 16 h)],  every reader (ka, kb) takes its 16 values c
  (odd strides 17 / 545 u64: conflict-free for the lanes along c, ka)

HALF mode (the reference's zero-padded forward transform of 2 Lh points, cuhe/Base.cu:309-437): the outputs of parity h
are the Lh-point transform of x[j] W^(j h), W = w_(2 Lh); W^(a T) = 2^(3 a) is a shift applied to the samples, W^m joins
the stage-1 twiddle, and Y[k] lands at X[2 k + h]."""

P = 0xFFFFFFFF00000001
G = 15893793146607301539          # cuhe/Base.cu:65: a primitive 65536-th root of unity


def root(length):
    return pow(G, 65536 // length, P)


_pw = {}


def dft(vals, w):
    n = len(vals)
    if (w, n) not in _pw:
        _pw[(w, n)] = [pow(w, e, P) for e in range(n)]
    pw = _pw[(w, n)]
    return [sum(vals[j] * pw[(j * k) % n] for j in range(n)) % P for k in range(n)]


def x1_addr(R, ka_l, c, b):
    return (ka_l * 32 + c) * (R + 1) + b


def x2_addr(kb_l, ka, c):
    return (kb_l * 32 + ka) * 33 + c


def lds_words(R):
    return max(16 * 32 * (R + 1), (R // 2) * 32 * 33)


def simulate(R, u, half=False, h=0):
    """u: list of Lh samples (already index-mapped: u[j] = sample j of the sub-transform, before any twist).
    Returns Y[0..Lh): the Lh-point transform (HALF: the parity-h outputs of the 2 Lh-point transform of u)."""
    T = 32 * R
    Lh = 32 * T
    NP = 32 // R
    w = root(Lh)
    W = root(2 * Lh)
    w32 = pow(2, 6, P)
    wR = pow(2, 192 // R, P)
    assert pow(w, T, P) == w32 and pow(w, 1024, P) == wR
    tw1 = lambda ka, m: pow(W, m * (2 * ka + 1), P) if (half and h) else pow(w, m * ka, P)
    tw2 = lambda kb, c: pow(w, 32 * c * kb, P)
    # ---- stage 1
    A = []
    for m in range(T):
        x = [u[a * T + m] for a in range(32)]
        if half and h:
            x = [x[a] * pow(2, 3 * a, P) % P for a in range(32)]
        a_ = dft(x, w32)
        A.append([a_[ka] * tw1(ka, m) % P for ka in range(32)])
    # ---- X1 in two halves through a buffer of lds_words(R)
    y = [[[None] * R for _ in range(NP)] for _ in range(T)]
    for hh in range(2):
        buf = [None] * lds_words(R)
        for m in range(T):
            b, c = m // 32, m % 32
            for ka in range(16 * hh, 16 * hh + 16):
                ad = x1_addr(R, ka - 16 * hh, c, b)
                assert buf[ad] is None
                buf[ad] = A[m][ka]
        for t2 in range(T):
            c, kq = t2 % 32, t2 // 32
            for i in range(NP):
                ka = kq + R * i
                if not (16 * hh <= ka < 16 * hh + 16):
                    continue
                for b in range(R):
                    y[t2][i][b] = buf[x1_addr(R, ka - 16 * hh, c, b)]
    # ---- stage 2
    Bv = []
    for t2 in range(T):
        c = t2 % 32
        rows = []
        for i in range(NP):
            d = dft(y[t2][i], wR)
            rows.append([d[kb] * tw2(kb, c) % P for kb in range(R)])
        Bv.append(rows)
    # ---- X2 in two halves
    z = [[None] * 32 for _ in range(T)]
    for hh in range(2):
        buf = [None] * lds_words(R)
        lo = (R // 2) * hh
        for t2 in range(T):
            c, kq = t2 % 32, t2 // 32
            for i in range(NP):
                ka = kq + R * i
                for kb in range(lo, lo + R // 2):
                    ad = x2_addr(kb - lo, ka, c)
                    assert buf[ad] is None
                    buf[ad] = Bv[t2][i][kb]
        for t3 in range(T):
            ka, kb = t3 % 32, t3 // 32
            if not (lo <= kb < lo + R // 2):
                continue
            for c in range(32):
                z[t3][c] = buf[x2_addr(kb - lo, ka, c)]
    # ---- stage 3
    Y = [None] * Lh
    for t3 in range(T):
        d = dft(z[t3], w32)
        for kc in range(32):
            Y[t3 + T * kc] = d[kc]
    return Y


def x1_addr_bal(R, c, ka, b_l):
    RW = R // 2 + 1
    return c * (32 * RW + 1) + ka * RW + b_l


def x2_addr_bal(kb, ka, c_l):
    return kb * 544 + ka * 17 + c_l


def lds_words_bal(R):
    return max(32 * (32 * (R // 2 + 1) + 1), R * 544)


LDS_WORDS32 = lds_words_bal(32)


def simulate_balanced(R, u, half=False, h=0):
    """The exchanges of the persistent kernels (ntt_onewg.cuh: owb_*), R = 16 or 32: the read side is unconditional and balanced.
    Stage-2 thread t2 = kq + R c.
      X1, round hh: the waves with b in [hh R/2, (hh + 1) R/2) (t = 32 b + c) store all 32 A[ka] -> buf[c S1 + ka RW + (b - hh R/2)],
                    RW = R/2 + 1, S1 = 32 RW + 1; every reader (kq, c) takes its R/2 values b for each of its 32/R values ka = kq + R i
      X2, round hh: the waves with c in [16 hh, 16 hh + 16) store all 32 B[i][kb] -> buf[kb 544 + ka 17 + (c - 16 hh)];
                    every reader (ka, kb) takes its 16 values c
    Same contract as simulate(R, ...)."""
    T, Lh, NP, HB = 32 * R, 1024 * R, 32 // R, R // 2
    w, W = root(Lh), root(2 * Lh)
    w32, wR = pow(2, 6, P), pow(2, 192 // R, P)
    tw1 = lambda ka, m: pow(W, m * (2 * ka + 1), P) if (half and h) else pow(w, m * ka, P)
    tw2 = lambda kb, c: pow(w, 32 * c * kb, P)
    words = lds_words_bal(R)
    A = []
    for m in range(T):
        x = [u[a * T + m] for a in range(32)]
        if half and h:
            x = [x[a] * pow(2, 3 * a, P) % P for a in range(32)]
        a_ = dft(x, w32)
        A.append([a_[ka] * tw1(ka, m) % P for ka in range(32)])
    y = [[None] * 32 for _ in range(T)]
    for hh in range(2):
        buf = [None] * words
        for m in range(T):
            b, c = m // 32, m % 32
            if b // HB != hh:
                continue
            for ka in range(32):
                ad = x1_addr_bal(R, c, ka, b - HB * hh)
                assert buf[ad] is None
                buf[ad] = A[m][ka]
        for t2 in range(T):
            kq, c = t2 % R, t2 // R
            for i in range(NP):
                for bl in range(HB):
                    y[t2][i * R + HB * hh + bl] = buf[x1_addr_bal(R, c, kq + R * i, bl)]
    Bv = []
    for t2 in range(T):
        c = t2 // R
        row = []
        for i in range(NP):
            d = dft(y[t2][i * R:(i + 1) * R], wR)
            row.append([d[kb] * tw2(kb, c) % P for kb in range(R)])
        Bv.append(row)
    z = [[None] * 32 for _ in range(T)]
    for hh in range(2):
        buf = [None] * words
        for t2 in range(T):
            kq, c = t2 % R, t2 // R
            if c // 16 != hh:
                continue
            for i in range(NP):
                for kb in range(R):
                    ad = x2_addr_bal(kb, kq + R * i, c - 16 * hh)
                    assert buf[ad] is None
                    buf[ad] = Bv[t2][i][kb]
        for t3 in range(T):
            ka, kb = t3 % 32, t3 // 32
            for cl in range(16):
                z[t3][16 * hh + cl] = buf[x2_addr_bal(kb, ka, cl)]
    Y = [None] * Lh
    for t3 in range(T):
        d = dft(z[t3], w32)
        for kc in range(32):
            Y[t3 + T * kc] = d[kc]
    return Y


def simulate32(u, half=False, h=0):
    return simulate_balanced(32, u, half, h)
