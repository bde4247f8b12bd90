// ntt_onewg.cuh -- ONE-WORKGROUP transforms: a whole sub-transform of Lh = 4K / 8K / 16K / 32K points lives in the registers of
// one workgroup (32 values per thread, T = Lh/32 threads), three shift-only register stages 32 x R x 32 (R = 4 / 8 / 16 / 32),
// two general twiddle multiplications per point, two exchanges through LDS -- ONE launch, no slab in HBM: a transform's
// bytes cross the memory system once in and once out (the two-pass scheme of ntt_kernels.cuh writes and re-reads a
// u64[L] slab per transform, 2.6x the algorithmic traffic at 64K points).
//
// Replaces, like ntt_kernels.cuh, the reference's three-pass 64 x 64 x {4,8,16} scheme (cuhe/Base.cu:309-842,
// cuhe/Operations.cu:306-398).  The zero-padded forward transform of the reference contract (u32[L/2] -> u64[L],
// cuhe/Base.cu:309-437) is done as its two decimation-in-frequency halves: the outputs of parity h are the L/2-point
// transform of x[j] W^(j h), W = w_L -- W^(a T) = 2^(3a) is a shift of the samples, W^m joins the stage-1 twiddle table --
// one workgroup per half (HALF mode), both halves of a transform on one XCD.  The halves write alternate 8-byte words of the
// same lines: only stores issued close together in time meet in L2, which is why the persistent form lets the pair meet first.
//
// Dataflow (tests/onewg_model.py is the executable statement of the index formulas; tests/test_onewg_model.py pins it to
// the oracle):
//   stage 1  thread m:                  x[a] = u[a T + m];  A[ka] = DFT32_a(x) * TW1[ka T + m]         (w_Lh^(m ka))
//   X1       -> thread t2 = c + 32 kq:  y[i][b] = A_{m = 32 b + c}[ka],  ka = kq + R i, i < 32 / R
//   stage 2  B[i][kb] = DFT_R_b(y[i]) * TW2[32 kb + c]                                                 (w_T^(c kb))
//   X2       -> thread t3 = ka + 32 kb: z[c] = B_{(kq, c), i}[kb]
//   stage 3  Y[t3 + T kc] = DFT32_c(z), stored through the same epilogues as pass 2 (ntt_kernels.cuh: pass2_store)
// Each exchange moves half of every thread's values at a time (the LDS holds half a transform: 1 / 2 / 4 / 7 workgroups
// per CU at 32K / 16K / 8K / 4K points); rows are padded to odd strides (R + 1, 33 u64): conflict-free on both sides.  At
// R = 32 the exchanges use a layout whose read side is balanced and unconditional (ow32_*: stage-2 thread kq + 32 c).
//
// Two kernels share the stages: ntt_onewg (one workgroup per sub-transform: 2 / 4 of them overlap on a CU at 16K / 8K
// points) and ntt_onewg_stream (32K-point halves of the 64K-point zero-padded transform, where only ONE workgroup fits a
// CU: a persistent workgroup walks over its share of the batch, the u32 samples of the NEXT half arrive in the idle
// exchange buffer by LDS-DMA while stage 3 of the current one computes and stores, and the two workgroups of a row meet
// before their stores: 2.71-2.78 M transforms/s against 2.54-2.64 for the two-pass pair, 0.81 MB instead of 1.71 MB per
// transform at the L2/fabric boundary; profiles/r03_onewg_ab.txt).  The persistent kernel is generic in the sub-transform
// size (16K / 32K points) and also takes the full negacyclic rows of 64K points; with two workgroups per CU it loses to
// ntt_onewg (profiles/r03_split_rows.txt), so it serves the 64K-point rows only.
// Full-length negacyclic rows (the ciphertext domain) can run SPLIT the same way -- two half-length sub-transforms per row,
// both input halves present (see "SPLIT transforms" below): the default for the inverse rows of 32K points.
#pragma once
#include "ntt_kernels.cuh"

namespace cuhe {

constexpr int kOwGiveUpSlot = 512;          // = kOwPairCounters (ntt_onewg.hpp): the counter behind the rendezvous counters
template <int R>
struct OwGeom {
    static constexpr int T = 32 * R, Lh = 32 * T, NP = 32 / R;
    static constexpr int X1W = 16 * 32 * (R + 1), X2W = (R / 2) * 32 * 33;
    static constexpr int XWB = 32 * (32 * (R / 2 + 1) + 1) > R * 544 ? 32 * (32 * (R / 2 + 1) + 1) : R * 544;   // the balanced layouts (owb_*)
    static constexpr int XW = R == 32 ? XWB : (X1W > X2W ? X1W : X2W);            // exchange buffer (u64 words); R = 32: balanced
    static constexpr size_t bytes = (size_t)(XW + T) * sizeof(u64);   // + the stage-2 twiddle table
    static constexpr size_t bytes_stream = (size_t)(XWB + T) * sizeof(u64);       // persistent kernels: balanced at every size
};

// x * 2^K mod P for a sample below 2^32 (compile-time K < 96): nothing to reduce up to K = 32
template <int K>
__device__ __forceinline__ u64 shlp32(u32 x) {
    static_assert(K >= 0 && K < 96, "shift out of range");
    if constexpr (K <= 32) return (u64)x << K;                                   // < 2^64 - 2^32 < P
    else if constexpr (K < 64) return mad_eps(x >> (64 - K), (u64)x << K);
    else if constexpr (K == 64) return ((u64)x << 32) - x;                        // x * eps < P
    else {
        const u32 mid = x << (K - 64);                                            // bits 64..95
        const u64 t1 = ((u64)mid << 32) - mid;
        return subp(t1, (u64)(x >> (96 - K)));                                    // 2^96 = -1
    }
}
template <int A>
struct HalfShift {                                       // x[a] *= 2^(3a): the w_L^(a T h) factor of the odd half
    static __device__ __forceinline__ void run(u64 (&x)[32]) {
        x[A] = shlp32<3 * A>((u32)x[A]);
        if constexpr (A + 1 < 32) HalfShift<A + 1>::run(x);
    }
};

// the 32-point transform of stage 1 and its twiddle products.  The table values arrive in batches that are in flight during
// the butterflies (a first batch of 4: more would spill) / during the products of the previous batch (hipcc places a
// load next to its use: an L2 round trip per value, which one workgroup per CU has nothing to hide behind).
template <int T>
__device__ __forceinline__ void ow_dft_twiddle(u64 (&x)[32], const u64 *__restrict__ t1, bool row0) {
    u64 w0[4], wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = t1[i * T];
    __builtin_amdgcn_sched_barrier(0);
    dft_regs<32, false>(x);
#pragma unroll
    for (int i = 0; i < 8; ++i) wa[i] = t1[(4 + i) * T];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ka = 0; ka < 4; ++ka)
        if (ka != 0 || row0) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], w0[ka]);
#pragma unroll
    for (int i = 0; i < 8; ++i) wb[i] = t1[(12 + i) * T];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ka = 4; ka < 12; ++ka) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], wa[ka - 4]);
#pragma unroll
    for (int i = 0; i < 8; ++i) wa[i] = t1[(20 + i) * T];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ka = 12; ka < 20; ++ka) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], wb[ka - 12]);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = t1[(28 + i) * T];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ka = 20; ka < 28; ++ka) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], wa[ka - 20]);
#pragma unroll
    for (int ka = 28; ka < 32; ++ka) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], w0[ka - 28]);
}

// ---- stage 1 after the samples are in x: 32-point transform, stage-1 twiddles (t1 = table + m; row0: row ka = 0 of the
// table is not all ones), exchange 1.
template <int R>
__device__ __forceinline__ void ow_stage1_x1(u64 (&x)[32], u64 (&y)[32], u64 *buf, const u64 *__restrict__ t1, bool row0, int t) {
    using G = OwGeom<R>;
    constexpr int T = G::T, NP = G::NP;
    ow_dft_twiddle<T>(x, t1, row0);
    const int c = t & 31, q = t >> 5;                     // writer: (b, c) = (q, c); reader: (kq, c) = (q, c)
    {
        u64 *wr = buf + c * (R + 1) + q;
#pragma unroll
        for (int kl = 0; kl < 16; ++kl) wr[kl * 32 * (R + 1)] = x[bitrev<32>(kl)];
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if (hh == 1) {
            u64 *wr = buf + c * (R + 1) + q;
#pragma unroll
            for (int kl = 0; kl < 16; ++kl) wr[kl * 32 * (R + 1)] = x[bitrev<32>(16 + kl)];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            bool act;
            if constexpr (R == 32) act = (q >> 4) == hh;              // whole waves
            else act = ((R * i) >> 4) == hh;                          // known at compile time: ka = q + R i, q < R <= 16
            if (act) {
                const u64 *rd = buf + (((q + R * i) & 15) * 32 + c) * (R + 1);
#pragma unroll
                for (int b = 0; b < R; ++b) y[i * R + b] = rd[b];
            }
        }
        __syncthreads();
    }
}

// ---- stage 2 (NP transforms of R points, times w_T^(c kb)), exchange 2 (last_sync: a barrier after the last reads,
// for a caller that re-uses the buffer); the caller finishes with stage 3: dft_regs<32>(z), z[bitrev32(kc)] = Y[t + T kc]
template <int R>
__device__ __forceinline__ void ow_stage2_x2(u64 (&y)[32], u64 (&z)[32], u64 *buf, const u64 *tw2, int t, bool last_sync) {
    using G = OwGeom<R>;
    constexpr int NP = G::NP;
    const int c = t & 31, q = t >> 5;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        u64 (&sub)[R] = *reinterpret_cast<u64(*)[R]>(&y[i * R]);
        dft_regs<R, false>(sub);
#pragma unroll
        for (int kb = 1; kb < R; ++kb) sub[bitrev<R>(kb)] = mulp(sub[bitrev<R>(kb)], tw2[32 * kb + c]);
    }
    const int ka3 = t & 31, kb3 = t >> 5;                 // reader t3 = ka + 32 kb
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            u64 *w = buf + (q + R * i) * 33 + c;
#pragma unroll
            for (int kl = 0; kl < R / 2; ++kl) w[kl * 32 * 33] = y[i * R + bitrev<R>(hh * (R / 2) + kl)];
        }
        __syncthreads();
        if ((kb3 / (R / 2)) == hh) {                      // whole waves: t3 < T/2 or >= T/2
            const u64 *rd = buf + ((kb3 - hh * (R / 2)) * 32 + ka3) * 33;
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) z[cc] = rd[cc];
        }
        if (hh == 0 || last_sync) __syncthreads();
    }
}

// ---- exchanges whose READ side is unconditional and balanced (tests/onewg_model.py: simulate_balanced), R = 16 or 32: what
// the 32K-point transform (one workgroup of 1024 threads per CU) and the persistent kernels use.  Stage-2 thread t2 = kq + R c.
//   X1, round h: the waves with b in [h R/2, (h + 1) R/2) (t = 32 b + c) store all 32 A[ka] -> buf[c S1 + ka RW + (b - h R/2)],
//                RW = R/2 + 1, S1 = 32 RW + 1; every reader (kq, c) takes its R/2 values b of each of its 32/R values ka
//   X2, round h: the waves with c in [16 h, 16 h + 16) (t2 = kq + R c) store all 32 B[i][kb] -> buf[kb 544 + ka 17 + (c - 16 h)];
//                every reader (ka, kb) takes its 16 values c
// Every thread's y / z are assigned on every path, which also keeps them out of the loop-carried state of the persistent kernel.
template <int R>
__device__ __forceinline__ void owb_stage1_x1(u64 (&x)[32], u64 (&y)[32], u64 *buf, const u64 *__restrict__ t1, bool row0, int t) {
    static_assert(R == 16 || R == 32, "balanced exchanges: 16K / 32K points");
    constexpr int T = 32 * R, NP = 32 / R, HB = R / 2, RW = HB + 1, S1 = 32 * RW + 1;
    ow_dft_twiddle<T>(x, t1, row0);
    const int c1 = t & 31, b1 = t >> 5;                   // writer (b, c)
    const int kq = t % R, c2 = t / R;                     // reader (kq, c)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if ((b1 / HB) == hh) {                            // whole waves
            u64 *wr = buf + c1 * S1 + (b1 % HB);
#pragma unroll
            for (int ka = 0; ka < 32; ++ka) wr[ka * RW] = x[bitrev<32>(ka)];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const u64 *rd = buf + c2 * S1 + (kq + R * i) * RW;
#pragma unroll
            for (int bl = 0; bl < HB; ++bl) y[i * R + HB * hh + bl] = rd[bl];
        }
        __syncthreads();
    }
}
template <int R>
__device__ __forceinline__ void owb_stage2_x2(u64 (&y)[32], u64 (&z)[32], u64 *buf, const u64 *tw2, int t, bool last_sync) {
    static_assert(R == 16 || R == 32, "balanced exchanges: 16K / 32K points");
    constexpr int NP = 32 / R;
    const int kq = t % R, c = t / R;                      // stage 2: (kq, c); stage 3: (ka, kb) = (t & 31, t >> 5)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        u64 (&sub)[R] = *reinterpret_cast<u64(*)[R]>(&y[i * R]);
        dft_regs<R, false>(sub);
#pragma unroll
        for (int kb = 1; kb < R; ++kb) sub[bitrev<R>(kb)] = mulp(sub[bitrev<R>(kb)], tw2[32 * kb + c]);
    }
    const int ka3 = t & 31, kb3 = t >> 5;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if ((c >> 4) == hh) {                             // whole waves
            u64 *wr = buf + kq * 17 + (c & 15);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int kb = 0; kb < R; ++kb) wr[kb * 544 + (R * i) * 17] = y[i * R + bitrev<R>(kb)];
        }
        __syncthreads();
        const u64 *rd = buf + kb3 * 544 + ka3 * 17;
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) z[16 * hh + cl] = rd[cl];
        if (hh == 0 || last_sync) __syncthreads();
    }
}

// the outputs of parity h of the zero-padded transform: Y[k] -> X[2 k + h]
template <int R, int OUT>
__device__ __forceinline__ void ow_store_half(const u64 (&z)[32], void *dst_, long dst_stride, int batch, int h, int t,
                                              const u64 *__restrict__ xtab, int prime0, int np_mod) {
    constexpr int T = OwGeom<R>::T;
    constexpr long L = 2L * OwGeom<R>::Lh;
    static_assert(OUT == kOutU64 || OUT == kOutU64Mul, "HALF mode stores u64 rows");
    u64 *dst = (u64 *)dst_ + (long)batch * dst_stride + h;
    if constexpr (OUT == kOutU64Mul) {
        const int pidx = np_mod > 0 ? (prime0 + batch) % np_mod : prime0 + batch;
        const u64 *tab = xtab + (long)pidx * L + h;
#pragma unroll
        for (int kc = 0; kc < 32; ++kc) { const long o = 2L * (t + T * kc); dst[o] = mulp(z[bitrev<32>(kc)], tab[o]); }
    } else {
#pragma unroll
        for (int kc = 0; kc < 32; ++kc) dst[2L * (t + T * kc)] = z[bitrev<32>(kc)];
    }
}

// ---- SPLIT transforms: a full-length row of L = 2 Lh points done as the two Lh-point transforms of its even and odd outputs,
// one workgroup each (decimation in frequency, like the zero-padded form, but both input halves are there).
// Forward NEGACYCLIC (the ciphertext domain of x^L + 1; psi the primitive 2L-th root, psi^2 = w_L, i4 = psi^Lh = +-2^48):
//   X[2k + h] = sum_{j < Lh} u_h[j] w_Lh^(j k),   u_h[j] = (x[j] + (-1)^h i4 x[j + Lh]) psi^((1 + 2h) j).
// With j = T a + m: psi^((1 + 2h) T a) = c^((1 + 2h) a), c = psi^T a 128-th root with c^2 = 8 -- a shift, times c when the
// exponent is odd (16 of the 32 samples of a thread, in BOTH halves: equal work) -- and psi^((1 + 2h) m) joins the stage-1
// table, TW1g[h][ka T + m] = psi^(m (1 + 2h + 4 ka)).
// Inverse negacyclic (u64 rows Xs, or the products of two rows, -> u32 coefficients): with Y[k] = Xs[(L - k) mod L]
//   x[2j + h] = psi^-(2j + h) / L * sum_{k < Lh} v_h[k] w_Lh^(j k),   v_h[k] = (Y[k] + (-1)^h Y[k + Lh]) W^(h k),  W = w_L:
// W^(T a) = 2^(3a) is a shift, W^m and 1 / L join the stage-1 table (TW1hi = the parity tables of the zero-padded form / L),
// psi^-(2j + h), the centred lift and the reduction modulo p_i are the store epilogue of kOutModPNc at index 2j + h.
template <int K>
__device__ __forceinline__ u64 mulpow2(u64 v) {           // v * 2^K, K in [0, 192)
    if constexpr (K >= 96) return negp(shlp<K - 96>(v));
    else return shlp<K>(v);
}
template <int H, int A>
struct StreamTwist {                                      // x[a] *= c^((1 + 2H) a)
    static __device__ __forceinline__ void run(u64 (&x)[32], u64 c) {
        constexpr int e = (1 + 2 * H) * A;
        u64 v = mulpow2<(3 * (e >> 1)) % 192>(x[A]);
        if constexpr (e & 1) v = mulp(v, c);
        x[A] = v;
        if constexpr (A + 1 < 32) StreamTwist<H, A + 1>::run(x, c);
    }
};
struct StreamTwistArgs { u64 c128; int i4neg; };            // c = psi^1024; i4neg: psi^32768 = -2^48 (else +2^48)

template <int A>
struct SplitShift {                                      // x[a] *= 2^(3a) for full-width values: the W^(a T) factor of the odd half
    static __device__ __forceinline__ void run(u64 (&x)[32]) {
        if constexpr (A > 0) x[A] = shlp<3 * A>(x[A]);
        if constexpr (A + 1 < 32) SplitShift<A + 1>::run(x);
    }
};
// the outputs of parity h of the inverse negacyclic transform: Y[j] -> coefficient 2 j + h (kOutModPNc, see pass2_store)
template <int R>
__device__ __forceinline__ void ow_store_half_nc(const u64 (&z)[32], void *dst_, long dst_stride, int batch, int h, int t,
                                                 const u64 *__restrict__ xtab, u32 p, u64 m) {
    constexpr int T = OwGeom<R>::T;
    u32 *dst = (u32 *)dst_ + (long)batch * dst_stride + h;
    const u64 *ti = xtab + h;
#pragma unroll
    for (int kc = 0; kc < 32; ++kc) {
        const long o = 2L * (t + T * kc);
        const u64 v = mulp(z[bitrev<32>(kc)], ti[o]);
        const bool neg = v > (kP >> 1);                       // centred lift: v - P < 0
        const u32 rr = mod_small(neg ? kP - v : v, p, m);
        dst[o] = (neg && rr) ? p - rr : rr;
    }
}

// RowRebase (2 KB) travels BY VALUE in the kernel-argument segment of every launch, list or not: the GPU reads only `per` (one scalar
// load) unless the launch is a list, the host copies 2 KB more per dispatch.  ADVICE r05 asked for the A/B: -DCUHE_OW_NO_REBASE_ARG builds
// the kernels without the argument (list launches refused); launch-bound calls of 32 rows and the headline batch, both builds
// alternating on one box: profiles/r06_rebase_arg_ab.txt.
#ifdef CUHE_OW_NO_REBASE_ARG
#define OW_RB_PARAM
#else
#define OW_RB_PARAM , RowRebase rb
#endif
// LGH: log2 of the sub-transform; HALF: the transform has 2^(LGH+1) points and this workgroup produces the outputs of one
// parity -- of the zero-padded forward transform (a source with a zero upper half) or of a SPLIT full-length row (above).
// TW1: HALF ? u64[2][Lh] (parity h at + h Lh) : u64[Lh].
template <int LGH, int MODE, int OUT, bool HALF>
__global__ __launch_bounds__(OwGeom<(1 << LGH) / 1024>::T, 4)
void ntt_onewg(void *__restrict__ dst_, const void *__restrict__ src_, const u64 *__restrict__ TW1, const u64 *__restrict__ TW2,
               long src_stride, long dst_stride, int nbatch, int nstore, WindowArgs wa, const u64 *__restrict__ tw,
               const u32 *__restrict__ primes, const u64 *__restrict__ pinv, int prime0, int np_mod,
               const u32 *__restrict__ aux, long aux_stride, FoldGeom fg, const u64 *__restrict__ xtab, StreamTwistArgs ta OW_RB_PARAM) {
    constexpr int R = (1 << LGH) / 1024;
    using G = OwGeom<R>;
    constexpr int T = G::T, Lh = G::Lh;
    constexpr int LGF = HALF ? LGH + 1 : LGH;             // log2 of the transform the caller sees
    constexpr bool INV = out_is_inverse(OUT);
    constexpr bool SPLIT = HALF && !src_is_ext(MODE);
    constexpr bool SPLIT_FWD = SPLIT && MODE == kSrcU32Twist, SPLIT_INV = SPLIT && (MODE == kSrcU64Neg || MODE == kSrcU64NegMul);
    static_assert(!HALF || SPLIT || (src_is_ext(MODE) && !INV), "HALF mode of a zero-padded source is the forward transform");
    static_assert(!SPLIT || (SPLIT_FWD && (OUT == kOutU64 || OUT == kOutU64Mul)) || (SPLIT_INV && OUT == kOutModPNc), "split rows: the negacyclic pair");
    static_assert(HALF || !src_is_ext(MODE), "a zero-padded source goes through HALF mode");
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 *buf = lds;
    u64 *tw2 = lds + G::XW;

    int batch, h = 0;
    if constexpr (HALF) {
        const int g = blockIdx.x, r = g >> 3;             // blocks g and g + 8: the two halves of one transform, same XCD
        h = r & 1;
        batch = (r >> 1) * 8 + (g & 7);
    } else batch = blockIdx.x;
    if (batch >= nbatch) return;
#ifndef CUHE_OW_NO_REBASE_ARG
    if (rb.per > 0) {                                     // rows in separate blocks: the block of this row (uniform in the workgroup)
        const int c = batch / rb.per;
        src_ = (const char *)src_ + rb.src_adj[c];
        dst_ = (char *)dst_ + rb.dst_adj[c];
    }
#endif
    const int t = threadIdx.x;
    tw2[t] = TW2[t];

    u64 x[32], y[32], z[32];
    if constexpr (SPLIT_FWD) {
        const u32 *row = (const u32 *)src_ + (long)batch * src_stride + t;
        const bool neg = ((h ^ ta.i4neg) & 1) != 0;       // the sign of (-1)^h i4 / 2^48
#pragma unroll
        for (int a0 = 0; a0 < 32; a0 += 8) {
            u32 xl[8], xh[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) { xl[a] = row[(a0 + a) * T]; xh[a] = row[(a0 + a) * T + Lh]; }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                u64 sft = shlp32<48>(xh[a]);
                if (neg) sft = negp(sft);
                x[a0 + a] = addp((u64)xl[a], sft);
            }
        }
        if (h) StreamTwist<1, 0>::run(x, ta.c128); else StreamTwist<0, 0>::run(x, ta.c128);
    } else if constexpr (SPLIT_INV) {
        constexpr int L = 2 * Lh;
        const long ro = (long)batch * src_stride;
        const u64 *row = (const u64 *)src_ + ro;
#pragma unroll
        for (int a0 = 0; a0 < 32; a0 += 4) {
            u64 y0[4], y1[4], w0[4], w1[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int k = (a0 + a) * T + t;
                y0[a] = row[(L - k) & (L - 1)]; y1[a] = row[Lh - k];
                if constexpr (MODE == kSrcU64NegMul) { w0[a] = tw[ro + ((L - k) & (L - 1))]; w1[a] = tw[ro + Lh - k]; }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if constexpr (MODE == kSrcU64NegMul) { y0[a] = mulp(y0[a], w0[a]); y1[a] = mulp(y1[a], w1[a]); }
                x[a0 + a] = h ? subp(y0[a], y1[a]) : addp(y0[a], y1[a]);
            }
        }
        if (h) SplitShift<0>::run(x);
    } else {
#pragma unroll
        for (int a = 0; a < 32; ++a) x[a] = load_sample<LGF, MODE>(src_, src_stride, batch, a * T + t, wa, tw);
        if constexpr (HALF) { if (h) HalfShift<0>::run(x); }
    }
    if constexpr (SPLIT) {
        // (both parity tables carry a factor on every row: psi^((1 + 2h) m) or W^(h m) / L)
        if constexpr (R == 32) { owb_stage1_x1<32>(x, y, buf, TW1 + (long)h * Lh + t, true, t); owb_stage2_x2<32>(y, z, buf, tw2, t, false); }
        else { ow_stage1_x1<R>(x, y, buf, TW1 + (long)h * Lh + t, true, t); ow_stage2_x2<R>(y, z, buf, tw2, t, false); }
    } else if constexpr (R == 32) {
        owb_stage1_x1<32>(x, y, buf, TW1 + (HALF ? (long)h * Lh : 0) + t, INV || (HALF && h), t);
        owb_stage2_x2<32>(y, z, buf, tw2, t, false);
    } else {
        ow_stage1_x1<R>(x, y, buf, TW1 + (HALF ? (long)h * Lh : 0) + t, INV || (HALF && h), t);
        ow_stage2_x2<R>(y, z, buf, tw2, t, false);
    }
    dft_regs<32, false>(z);
    if constexpr (SPLIT_INV) {
        const int pidx = np_mod > 0 ? (prime0 + batch) % np_mod : prime0 + batch;
        ow_store_half_nc<R>(z, dst_, dst_stride, batch, h, t, xtab, primes[pidx], pinv[pidx]);
    } else if constexpr (HALF) {
        ow_store_half<R, OUT>(z, dst_, dst_stride, batch, h, t, xtab, prime0, np_mod);
    } else {
        // the pass-2 epilogues take the four outputs X[k1 + N1 (b + 16 cc)], N1 = Lh / 64 = T / 2:  t3 = k1 + N1 hi,
        // kc = b' + 8 cc  <=>  b = hi + 2 b'
        constexpr int N1 = T / 2;
        const int k1 = t & (N1 - 1), hi = t / N1;
        const P2Store A = pass2_store_args<LGH, OUT>(dst_, dst_stride, nstore, primes, pinv, prime0, np_mod, aux, aux_stride, fg, xtab, batch);
#pragma unroll
        for (int bp = 0; bp < 8; ++bp) {
            u64 y4[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) y4[bitrev<4>(cc)] = z[bitrev<32>(bp + 8 * cc)];
            pass2_store<LGH, OUT>(y4, hi + 2 * bp, k1, batch, A);
        }
    }
}

// ---- persistent form for the 32K-point halves of the 64K-point zero-padded forward transform (u32 rows).
// Work item i = 2 * transform + parity, in the order of ntt_onewg's blocks; workgroup g takes items g, g + grid, ...
// The u32 samples of an item (128 KB) arrive by LDS-DMA in the exchange buffer, which is idle from the last read of
// exchange 2 to the first write of exchange 1 of the next item: the transfer runs beside stage 3 and its stores.
__device__ __forceinline__ void glds16(const void *gsrc, u32 lds_byte_addr) {     // 64 lanes x 16 B -> LDS [addr, addr + 1 KB), lane-linear
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// H: the parity this workgroup produces.  The grid is a multiple of 16 workgroups, so that the parity (item >> 3) & 1 is the
// same for every item of a workgroup and the odd half's sample shifts sit on a straight-line path.
// SRC = kSrcU32Ext: the zero-padded transform (32K samples per row); kSrcU32Twist: the negacyclic transform of a full row of 64K
// samples -- the lower 32K by LDS-DMA like the zero-padded form, the upper 32K straight from global memory, requested before
// the wait for the DMA.
template <int R, int SRC, int OUT, int H>
__device__ __forceinline__ void ow_stream_loop(void *__restrict__ dst_, const u32 *__restrict__ src, const u64 *__restrict__ TW1, u64 *buf, const u64 *tw2,
                                               long src_stride, long dst_stride, int nbatch, const u64 *__restrict__ xtab, int prime0, int np_mod,
                                               unsigned *pair_cnt, int *give_up, StreamTwistArgs ta) {
    static_assert(SRC == kSrcU32Ext || SRC == kSrcU32Twist, "row sources of the persistent form");
    static_assert(R == 16 || R == 32, "sub-transforms of 16K / 32K points");
    using G = OwGeom<R>;
    constexpr int T = G::T, Lh = G::Lh;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int nitems = 2 * ((nbatch + 7) & ~7);
    const u32 lds_base = (u32)(uintptr_t)buf;              // LDS byte address of the buffer (generic -> local: low 32 bits)
    unsigned *const gave_up_total = pair_cnt ? pair_cnt + kOwGiveUpSlot : nullptr;
    // items of a padding transform (batch rounded up to 8) are computed on the last real row and not stored: no divergent
    // control flow around the barriers
    auto fetch = [&](int i) {                              // samples of item i -> buf (as u32[Lh]); wave w moves bytes [8 KB w, 8 KB (w+1))
        const int b = min((i >> 4) * 8 + (i & 7), nbatch - 1);
        const char *row = (const char *)(src + (long)b * src_stride) + lane * 16;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const u32 off = (u32)(wave * 8 + p) * 1024u;
            glds16(row + off, lds_base + off);
        }
    };
    if ((int)blockIdx.x < nitems) fetch(blockIdx.x);
    int round = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int batch = (item >> 4) * 8 + (item & 7);
        // an offset the compiler cannot see through, new in every iteration: otherwise the ~100 loop-invariant table and
        // LDS addresses of the body are hoisted out of the loop and live across it (600+ bytes of spills per lane)
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        u32 xh[32];
        if constexpr (SRC == kSrcU32Twist) {
            const u32 *hi = src + (long)min(batch, nbatch - 1) * src_stride + Lh + t + opaque;
#pragma unroll
            for (int a = 0; a < 32; ++a) xh[a] = hi[a * T];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // every wave's part of the samples has landed
        u64 *lb = buf + opaque;
        u64 x[32], y[32], z[32];
        {
            const u32 *in = (const u32 *)lb + t;
#pragma unroll
            for (int a = 0; a < 32; ++a) x[a] = in[a * T];
        }
        __syncthreads();                                   // samples are in registers: the buffer is free for exchange 1
        if constexpr (SRC == kSrcU32Twist) {
            const bool neg = ((H ^ ta.i4neg) & 1) != 0;   // the sign of (-1)^h i4 / 2^48
#pragma unroll
            for (int a = 0; a < 32; ++a) {
                u64 sft = shlp32<48>(xh[a]);
                if (neg) sft = negp(sft);
                x[a] = addp(x[a], sft);
            }
            StreamTwist<H, 0>::run(x, ta.c128);
        } else if constexpr (H) HalfShift<0>::run(x);
        owb_stage1_x1<R>(x, y, lb, TW1 + (long)H * Lh + t + opaque, H != 0 || SRC == kSrcU32Twist, t);
        owb_stage2_x2<R>(y, z, lb, tw2 + opaque, t, true);
        if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);      // the buffer is idle until exchange 1 of the next item
        // Rendezvous with the workgroup that computes the OTHER parity of this row (block ^ 8: same XCD, same round): the two
        // write alternate 8-byte words of the same lines, and only stores issued within a few microseconds of each other meet in
        // L2 (measured: 1.9x the output bytes leave L2 when the pair drifts -- the odd half has 3 % more work --, 2.33 -> 2.72 M
        // transforms/s when it does not; tools/ubench_onewg.hip).  Performance only: the wait is bounded and nothing depends on
        // it having succeeded.  Every workgroup of the grid is resident (one per CU), so the partner is running.
        if (pair_cnt) {
            if (t == 0) {
                unsigned *c = pair_cnt + (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7));
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = 2u * (unsigned)(round + 1);
                bool met = false;
                for (int spin = 0; spin < 256 && !met; ++spin) {
                    met = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
                    if (!met) __builtin_amdgcn_s_sleep(4);
                }
                if (!met) {                                 // the partner is not running beside us (another kernel holds its CU): stop waiting for it
                    *give_up = 1;
                    __hip_atomic_fetch_add(gave_up_total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // visible to the host: cuhe_hip_last_dispatch_info
                }
            }
            __syncthreads();
            if (*give_up) pair_cnt = nullptr;
        }
        dft_regs<32, false>(z);
        if (batch < nbatch) ow_store_half<R, OUT>(z, dst_, dst_stride, batch, H, t, xtab, prime0, np_mod);
        ++round;
    }
}
template <int LGH, int SRC, int OUT>
__global__ __launch_bounds__(OwGeom<(1 << LGH) / 1024>::T, 4)
void ntt_onewg_stream(void *__restrict__ dst_, const u32 *__restrict__ src, const u64 *__restrict__ TW1, const u64 *__restrict__ TW2,
                      long src_stride, long dst_stride, int nbatch, const u64 *__restrict__ xtab, int prime0, int np_mod, unsigned *pair_cnt,
                      StreamTwistArgs ta) {
    constexpr int R = (1 << LGH) / 1024;
    using G = OwGeom<R>;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 *buf = lds;
    u64 *tw2 = lds + G::XWB;
    __shared__ int give_up;
    tw2[threadIdx.x] = TW2[threadIdx.x];
    if (threadIdx.x == 0) give_up = 0;
    if ((blockIdx.x >> 3) & 1) ow_stream_loop<R, SRC, OUT, 1>(dst_, src, TW1, buf, tw2, src_stride, dst_stride, nbatch, xtab, prime0, np_mod, pair_cnt, &give_up, ta);
    else ow_stream_loop<R, SRC, OUT, 0>(dst_, src, TW1, buf, tw2, src_stride, dst_stride, nbatch, xtab, prime0, np_mod, pair_cnt, &give_up, ta);
}

// (Round 5 built the persistent form of the SPLIT inverse of 64K-point rows -- resident 32K-point workgroups, rendezvous of a row's two
// halves before the stores of coefficient 2 j + h -- and removed it: 1.82 ms against 1.78 ms for the two-pass pair per 32 ciphertexts of
// x^65536+1; profiles/r05_split_inverse_ab.txt.)


}  // namespace cuhe
