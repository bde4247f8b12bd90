// ntt_onewg.cuh -- ONE-WORKGROUP transforms: a whole sub-transform of Lh = 8K / 16K / 32K points lives in the registers of
// one workgroup (32 values per thread, T = Lh/32 threads), three shift-only register stages 32 x R x 32 (R = 8 / 16 / 32),
// two general twiddle multiplications per point, two exchanges through LDS -- ONE launch, no slab in HBM: a transform's
// bytes cross the memory system once in and once out (the two-pass scheme of ntt_kernels.cuh writes and re-reads a
// u64[L] slab per transform, 2.6x the algorithmic traffic at 64K points).
//
// Replaces, like ntt_kernels.cuh, the reference's three-pass 64 x 64 x {4,8,16} scheme (cuhe/Base.cu:309-842,
// cuhe/Operations.cu:306-398).  The zero-padded forward transform of the reference contract (u32[L/2] -> u64[L],
// cuhe/Base.cu:309-437) is done as its two decimation-in-frequency halves: the outputs of parity h are the L/2-point
// transform of x[j] W^(j h), W = w_L -- W^(a T) = 2^(3a) is a shift of the samples, W^m joins the stage-1 twiddle table --
// one workgroup per half (HALF mode), both halves of a transform on one XCD so that their interleaved 8-byte stores
// meet in its L2.
//
// Dataflow (tests/onewg_model.py is the executable statement of the index formulas; tests/test_onewg_model.py pins it to
// the oracle):
//   stage 1  thread m:                  x[a] = u[a T + m];  A[ka] = DFT32_a(x) * TW1[ka T + m]         (w_Lh^(m ka))
//   X1       -> thread t2 = c + 32 kq:  y[i][b] = A_{m = 32 b + c}[ka],  ka = kq + R i, i < 32 / R
//   stage 2  B[i][kb] = DFT_R_b(y[i]) * TW2[32 kb + c]                                                 (w_T^(c kb))
//   X2       -> thread t3 = ka + 32 kb: z[c] = B_{(kq, c), i}[kb]
//   stage 3  Y[t3 + T kc] = DFT32_c(z), stored through the same epilogues as pass 2 (ntt_kernels.cuh: pass2_store)
// Each exchange moves half of every thread's values at a time (the LDS holds half a transform: 1 / 2 / 4 workgroups
// per CU at 32K / 16K / 8K points); rows are padded to odd strides (R + 1, 33 u64): conflict-free on both sides.
#pragma once
#include "ntt_kernels.cuh"

namespace cuhe {

template <int R>
struct OwGeom {
    static constexpr int T = 32 * R, Lh = 32 * T, NP = 32 / R;
    static constexpr int X1W = 16 * 32 * (R + 1), X2W = (R / 2) * 32 * 33;
    static constexpr int XW = X1W > X2W ? X1W : X2W;              // exchange buffer (u64 words)
    static constexpr size_t bytes = (size_t)(XW + T) * sizeof(u64);   // + the stage-2 twiddle table
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

// LGH: log2 of the sub-transform; HALF: the transform has 2^(LGH+1) points with a zero upper input half and this
// workgroup produces the outputs of one parity.  TW1: HALF ? u64[2][Lh] (parity h at + h Lh) : u64[Lh].
template <int LGH, int MODE, int OUT, bool HALF>
__global__ __launch_bounds__(OwGeom<(1 << LGH) / 1024>::T)
void ntt_onewg(void *__restrict__ dst_, const void *__restrict__ src_, const u64 *__restrict__ TW1, const u64 *__restrict__ TW2,
               long src_stride, long dst_stride, int nbatch, int nstore, WindowArgs wa, const u64 *__restrict__ tw,
               const u32 *__restrict__ primes, const u64 *__restrict__ pinv, int prime0, int np_mod,
               const u32 *__restrict__ aux, long aux_stride, FoldGeom fg, const u64 *__restrict__ xtab) {
    constexpr int R = (1 << LGH) / 1024;
    using G = OwGeom<R>;
    constexpr int T = G::T, Lh = G::Lh, NP = G::NP;
    constexpr int LGF = HALF ? LGH + 1 : LGH;             // log2 of the transform the caller sees
    constexpr bool INV = out_is_inverse(OUT);
    static_assert(!HALF || (src_is_ext(MODE) && !INV), "HALF mode is the zero-padded forward transform");
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
    const int t = threadIdx.x;
    tw2[t] = TW2[t];

    // ---- stage 1: thread m = t
    u64 x[32];
#pragma unroll
    for (int a = 0; a < 32; ++a) x[a] = load_sample<LGF, MODE>(src_, src_stride, batch, a * T + t, wa, tw);
    if constexpr (HALF) { if (h) HalfShift<0>::run(x); }
    dft_regs<32, false>(x);
    {
        const u64 *t1 = TW1 + (HALF ? (long)h * Lh : 0) + t;
        const bool row0 = INV || (HALF && h);              // row ka = 0 of the table is not all ones
#pragma unroll
        for (int ka = 0; ka < 32; ++ka)
            if (ka != 0 || row0) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], t1[ka * T]);
    }
    // ---- exchange 1
    const int c = t & 31, q = t >> 5;                     // writer: (b, c) = (q, c); reader: (kq, c) = (q, c)
    u64 y[32];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        {
            u64 *w = buf + c * (R + 1) + q;
#pragma unroll
            for (int kl = 0; kl < 16; ++kl) w[kl * 32 * (R + 1)] = x[bitrev<32>(16 * hh + kl)];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int ka = q + R * i;
            if ((ka >> 4) == hh) {                        // R = 32: wave-uniform; R < 32: known at compile time (i)
                const u64 *rd = buf + ((ka & 15) * 32 + c) * (R + 1);
#pragma unroll
                for (int b = 0; b < R; ++b) y[i * R + b] = rd[b];
            }
        }
        __syncthreads();
    }
    // ---- stage 2: NP transforms of R points, times w_T^(c kb)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        u64 (&sub)[R] = *reinterpret_cast<u64(*)[R]>(&y[i * R]);
        dft_regs<R, false>(sub);
#pragma unroll
        for (int kb = 1; kb < R; ++kb) sub[bitrev<R>(kb)] = mulp(sub[bitrev<R>(kb)], tw2[32 * kb + c]);
    }
    // ---- exchange 2
    const int ka3 = t & 31, kb3 = t >> 5;                 // reader t3 = ka + 32 kb
    u64 z[32];
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
        if (hh == 0) __syncthreads();
    }
    // ---- stage 3
    dft_regs<32, false>(z);                               // z[bitrev32(kc)] = Y[t3 + T kc]
    if constexpr (HALF) {
        constexpr long L = 2L * Lh;
        u64 *dst = (u64 *)dst_ + (long)batch * dst_stride + h;
        if constexpr (OUT == kOutU64Mul) {
            const int pidx = np_mod > 0 ? (prime0 + batch) % np_mod : prime0 + batch;
            const u64 *tab = xtab + (long)pidx * L + h;
#pragma unroll
            for (int kc = 0; kc < 32; ++kc) { const long o = 2L * (t + T * kc); dst[o] = mulp(z[bitrev<32>(kc)], tab[o]); }
        } else {
            static_assert(OUT == kOutU64 || OUT == kOutU64Mul, "HALF mode stores u64 rows");
#pragma unroll
            for (int kc = 0; kc < 32; ++kc) dst[2L * (t + T * kc)] = z[bitrev<32>(kc)];
        }
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

}  // namespace cuhe
