// ntt_kernels.cuh -- batched length-L cyclic NTT / INTT over Z_P for gfx950
// (L = 2^LG, LG in {14,15,16}; root w_L = g^(65536/L), natural order in/out).
//
// Replaces the reference's 3-pass 64x64x{4,8,16} scheme, one prime per launch,
// 64-thread blocks (cuhe/Base.cu:309-842, cuhe/Operations.cu:306-398) with a
// 2-pass "four-step" split  L = N1 x 64:
//
//   pass 1:  for every column j2 < 64, an N1-point DFT over the stride-64 samples
//       x[64*j1 + j2], written to scratch[j2][k1] (coalesced).
//   pass 2:  for every k1, a 64-point DFT over j2 of scratch[.][k1] times the outer
//       twiddle w_L^(j2*k1), stored as X[k1 + N1*k2] in natural order.
//
// Every sub-transform of <= 64 points is shift-only (8 = 2^3 is a 64-th root of unity
// mod P); exactly two general modular multiplications per point remain.  Both kernels are
// "wave-split" (ntt_pass1w / ntt_pass2w): 16 values per thread in every register stage,
// 64-point stages done as 16 x 4 over wave-uniform quarters through LDS.
//
// The inverse transform reuses the same passes on index-negated input
// (cuhe/Base.cu:454,622,799) with L^-1 folded into the outer twiddle table and
// `mod p_i` + u64->u32 narrowing (and the reduction mod x^(L/2)+1 when it applies) fused
// into the pass-2 store (cuhe/Base.cu:469-490).  The relinearisation window extraction
// (cuhe/Base.cu:345-372) can be fused into the pass-1 load.
#pragma once
#include "modp.cuh"

namespace cuhe {

// kSrcU32Twist: NEGACYCLIC forward transform of a full-length u32 row: x[j] * psi^j on load (psi a primitive 2L-th root of
// unity, psi^2 = w_L; table `tw`), then the plain length-L cyclic transform: X[k] = sum_j x[j] psi^(j(2k+1)).  Products of
// such transforms are products modulo x^L + 1 -- no zero padding, no reduction step (cuhe/Operations.cu:460-501 vanishes).
// kSrcU64NegMul: inverse transform of the pointwise PRODUCT of two rows (the second operand's rows behind `tw`, same stride):
// the product never exists in memory (cAnd followed by n2c: cuhe/CuHE.cu:570-581 writes it and reads it back).
enum : int { kSrcU32Ext = 0, kSrcWindow = 1, kSrcU64Neg = 2, kSrcU32Twist = 3, kSrcU64NegMul = 4 };
__host__ __device__ constexpr bool src_is_ext(int mode) { return mode == kSrcU32Ext || mode == kSrcWindow; }

// blockIdx -> (batch, tile) with every tile of one transform on one XCD
// (block b runs on XCD b % 8: MI355X_MICROARCH "Workgroup dispatch"; speed only).
__device__ __forceinline__ void xcd_map(int tiles, int &batch, int &tile) {
    int g = blockIdx.x;
    int xcd = g & 7, r = g >> 3;
    tile = r % tiles;
    batch = (r / tiles) * 8 + xcd;
}

struct WindowArgs { int words, w, wid0; };
// rows that live in SEPARATE blocks, `per` consecutive rows in each (the ciphertexts of a batch of gates: cuhe_hip_ct_ntt_list /
// cuhe_hip_ct_intt_list): the kernel keeps addressing row r as base + r * stride, and block c = r / per comes with the byte offset that
// makes that land in its own block -- adj[c] = (block_c - base) - c * per * stride bytes.  per = 0: one array (every other caller)
constexpr int kRowBlocksMax = 128;           // (2 KB of kernel arguments; a layer of PRINCE S-boxes offers groups of up to 128 ciphertexts)
struct RowRebase { int per; long src_adj[kRowBlocksMax]; long dst_adj[kRowBlocksMax]; };

// sample `idx` of transform `batch` as pass 1 sees it, for every source kind (see the enum above)
template <int LG, int MODE>
__device__ __forceinline__ u64 load_sample(const void *__restrict__ src_, long src_stride, int batch, int idx, const WindowArgs &wa,
                                           const u64 *__restrict__ tw) {
    constexpr int L = 1 << LG;
    if constexpr (MODE == kSrcU32Ext) {
        const u32 *src = (const u32 *)src_ + (long)batch * src_stride;
        return src[idx];
    } else if constexpr (MODE == kSrcWindow) {
        // cuhe/Base.cu:361-371: w-bit window `wid` of a W-word coefficient
        const u32 *co = (const u32 *)src_ + (long)idx * wa.words;
        const int bit = wa.w * (wa.wid0 + batch);
        const int wi = bit >> 5;
        u64 sv = co[wi];
        if (wi + 1 < wa.words) sv |= (u64)co[wi + 1] << 32;
        sv >>= (bit & 31);
        return sv & (u64)((1u << wa.w) - 1u);
    } else if constexpr (MODE == kSrcU32Twist) {
        const u32 *src = (const u32 *)src_ + (long)batch * src_stride;
        return mulp_u32(tw[idx], src[idx]);
    } else if constexpr (MODE == kSrcU64NegMul) {
        const long o = (long)batch * src_stride + ((L - idx) & (L - 1));
        return mulp(((const u64 *)src_)[o], tw[o]);
    } else {
        const u64 *src = (const u64 *)src_ + (long)batch * src_stride;
        return src[(L - idx) & (L - 1)];
    }
}

// first DIF stage when the upper half of the input is zero (u + 0, (u - 0)*w^j)
template <int N, int J>
struct ExtStage {
    static __device__ __forceinline__ void run(u64 (&x)[N]) {
        x[J + N / 2] = shlp<(192 / N) * J>(x[J]);
        if constexpr (J + 1 < N / 2) ExtStage<N, J + 1>::run(x);
    }
};

template <int N, bool EXT>
__device__ __forceinline__ void dft_regs(u64 (&x)[N]) {
    if constexpr (EXT) {
        ExtStage<N, 0>::run(x);
        if constexpr (N > 2) DifAll<N, N / 2>::run(x);
    } else {
        DifAll<N, N>::run(x);
    }
}

// kOutU64Mul: forward transform whose outputs are multiplied by a table row on the way out
// (the pointwise product with a precomputed NTT-domain constant, fused: `pinv` carries the table, u64[prime][L])
// kOutModPRevQ / kOutFoldFinal: the two inverse transforms of the folded generic reduction
// (cuhe_transforms.hip: barrett_impl) with the elementwise step that follows each of them done in the store:
//   RevQ      : the first Kq coefficients come out REVERSED (q[t] = C[Kq-1-t]) and zero up to Lh/2 -- the quotient, ready
//               as input of the next forward transform;
//   FoldFinal : r[i] = (g mod (x^Lh - 1))[i] - (q Phi mod (x^Lh - 1))[i] for i < n, zero up to the row length, with g the
//               fold of the product row f (`aux`) modulo x^m - 1.
// kOutModPNc: inverse NEGACYCLIC transform: the outputs are multiplied by psi^-j (table `xtab`; L^-1 sits in the outer
//   twiddles as for every inverse), lifted to the centred representative (the integer coefficient of a product modulo
//   x^L + 1 lies in (-P/2, P/2) when 2 L p^2 < P) and reduced modulo p_i.
enum : int { kOutU64 = 0, kOutModP = 1, kOutModPFoldXn1 = 2, kOutU64Mul = 3, kOutModPRevQ = 4, kOutFoldFinal = 5, kOutModPNc = 6 };
__host__ __device__ constexpr bool out_is_inverse(int out) { return out == kOutModP || out == kOutModPFoldXn1 || out == kOutModPRevQ || out == kOutFoldFinal || out == kOutModPNc; }

// ---- folded form of the generic reduction.  f = product of two reduced polynomials (degree <= 2n-2, row stride nlen,
// residues < p).  g = f mod (x^m - 1) when m < 2n-1 (Phi_m divides x^m - 1), else g = f; D = length of g;
// Kq = D - n = length of the quotient q = floor(g / Phi); Lh = length of the half-length transforms.
struct FoldGeom { int n, m, D, Kq, Lh; };
__device__ __forceinline__ u32 fold_g(const u32 *row, int i, const FoldGeom &G, u32 p) {      // g[i], i < D
    u32 a = row[i];
    if (G.D == G.m && i + G.m <= 2 * G.n - 2) { a += row[i + G.m]; if (a >= p) a -= p; }
    return a;
}

// ---------------------------------------------------------------------------------------------------------------
// pass 2, wave-split form (ntt_pass2w): the 64-point DFT over j2 is done as 16 x 4 by FOUR waves of a 256-thread
// workgroup that owns 64 adjacent k1.  Wave r loads the 16 samples j2 = 4a + r (outer twiddle applied on load), does
// a 16-point DFT in registers, multiplies by w_64^(r*b) = 2^(3rb) -- r is wave-uniform, so each wave runs ONE
// specialised, divergence-free code path with compile-time shifts -- and writes A_r[b] to LDS.  After one barrier wave
// w reads, for its four b = 4i + w, the four A_r[b] and finishes with 4-point DFTs (w_4 = 2^48), storing
// X[k1 + N1*(b + 16c)].  A thread carries 16 values instead of 64: ~4x shorter critical path per wave, ~3x more
// resident waves (about 64 VGPRs, 32 KiB LDS per workgroup), which is what small batches (np <= 103 transforms per
// API call) and the load/store phases need; the arithmetic per point is about the same as the 64-in-registers form.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int kP2wCols = 64;                      // k1 per workgroup
static constexpr size_t kP2wLdsBytes = (size_t)64 * kP2wCols * sizeof(u64);

template <int R, int B>
struct TwShift {                                         // v * 2^(3*R*B) mod P, compile-time R, B
    static __device__ __forceinline__ u64 run(u64 v) {
        constexpr int K = (3 * R * B) % 192;
        if constexpr (K >= 96) return negp(shlp<K - 96>(v));
        else return shlp<K>(v);
    }
};
template <int R, int B>
struct StepAWrite {
    static __device__ __forceinline__ void run(const u64 (&x)[16], u64 *xch, int lane) {
        xch[(B * 4 + R) * kP2wCols + lane] = TwShift<R, B>::run(x[bitrev<16>(B)]);
        if constexpr (B + 1 < 16) StepAWrite<R, B + 1>::run(x, xch, lane);
    }
};

// Store epilogue of pass 2, shared by both forms of the kernel: the thread holds the four outputs X[k1 + N1*(b + 16c)],
// c < 4, of column k1 in y[bitrev4(c)] (what dft_regs<4> leaves) and applies what OUT asks for (see the enum above).
struct P2Store {
    void *dst_; long dst_stride; int nstore; const u32 *aux; long aux_stride; FoldGeom fg; const u64 *xtab;
    u32 p; u64 m; int pidx, k2full, rem;
};
template <int LG, int OUT>
__device__ __forceinline__ void pass2_store(const u64 (&y)[4], int b, int k1, int batch, const P2Store &A) {
    constexpr int L = 1 << LG, N1 = L / 64;
    const u32 p = A.p; const u64 m = A.m;
    if constexpr (OUT == kOutU64) {
        u64 *dst = (u64 *)A.dst_ + (long)batch * A.dst_stride + k1;
#pragma unroll
        for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(y[bitrev<4>(c)], &dst[(long)(b + 16 * c) * N1]);
    } else if constexpr (OUT == kOutU64Mul) {
        u64 *dst = (u64 *)A.dst_ + (long)batch * A.dst_stride + k1;
        const u64 *tab = A.xtab + (long)A.pidx * L + k1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long o = (long)(b + 16 * c) * N1;
            __builtin_nontemporal_store(mulp(y[bitrev<4>(c)], tab[o]), &dst[o]);
        }
    } else if constexpr (OUT == kOutModP) {
        u32 *dst = (u32 *)A.dst_ + (long)batch * A.dst_stride + k1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k2 = b + 16 * c;
            if (k2 < A.k2full || (k2 == A.k2full && k1 < A.rem)) dst[(long)k2 * N1] = mod_small(y[bitrev<4>(c)], p, m);
        }
    } else if constexpr (OUT == kOutModPNc) {
        u32 *dst = (u32 *)A.dst_ + (long)batch * A.dst_stride + k1;
        const u64 *ti = A.xtab + k1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long o = (long)(b + 16 * c) * N1;
            const u64 v = mulp(y[bitrev<4>(c)], ti[o]);
            const bool neg = v > (kP >> 1);                       // centred lift: v - P < 0
            const u32 rr = mod_small(neg ? kP - v : v, p, m);
            dst[o] = (neg && rr) ? p - rr : rr;
        }
    } else if constexpr (OUT == kOutModPRevQ) {
        u32 *dst = (u32 *)A.dst_ + (long)batch * A.dst_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = (b + 16 * c) * N1 + k1;
            if (idx < A.fg.Kq) dst[A.fg.Kq - 1 - idx] = mod_small(y[bitrev<4>(c)], p, m);
            else if (idx < A.nstore) dst[idx] = 0u;
        }
    } else if constexpr (OUT == kOutFoldFinal) {
        u32 *dst = (u32 *)A.dst_ + (long)batch * A.dst_stride;
        const u32 *frow = A.aux + (long)batch * A.aux_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int idx = (b + 16 * c) * N1 + k1;
            if (idx < A.fg.n) {
                u32 a = fold_g(frow, idx, A.fg, p);
                if (idx + A.fg.Lh < A.fg.D) { a += fold_g(frow, idx + A.fg.Lh, A.fg, p); if (a >= p) a -= p; }
                const u32 qphi = mod_small(y[bitrev<4>(c)], p, m);
                dst[idx] = a >= qphi ? a - qphi : a + p - qphi;
            } else if (idx < A.nstore) dst[idx] = 0u;
        }
    } else {
        // inverse transform of a product fused with the reduction modulo x^(L/2) + 1 (inttMod when Phi_m = x^n + 1,
        // n = L/2): the thread owns f[i] (k2 = b + 16c) and f[i + n] (k2 + 32, i.e. c + 2); r[i] = (f[i] - f[i+n]) mod p_i.
        // The values are the exact integer coefficients (< P), so the signed difference is reduced once.
        u32 *dst = (u32 *)A.dst_ + (long)batch * A.dst_stride + k1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u64 a = y[bitrev<4>(c)], bb = y[bitrev<4>(c + 2)];
            const bool neg = a < bb;
            const u32 rr = mod_small(neg ? bb - a : a - bb, p, m);
            dst[(long)(b + 16 * c) * N1] = (neg && rr) ? p - rr : rr;
        }
    }
}
template <int LG, int OUT>
__device__ __forceinline__ P2Store pass2_store_args(void *dst_, long dst_stride, int nstore, const u32 *primes, const u64 *pinv, int prime0, int np_mod,
                                                    const u32 *aux, long aux_stride, const FoldGeom &fg, const u64 *xtab, int batch) {
    constexpr int N1 = (1 << LG) / 64;
    constexpr bool INV = out_is_inverse(OUT);
    P2Store A{dst_, dst_stride, nstore, aux, aux_stride, fg, xtab, 0u, 0ull, 0, 64, 0};
    A.pidx = np_mod > 0 ? (prime0 + batch) % np_mod : prime0 + batch;
    if constexpr (INV) { A.p = primes[A.pidx]; A.m = pinv[A.pidx]; A.k2full = nstore / N1; A.rem = nstore % N1; }
    return A;
}

template <int LG, int OUT>
__global__ __launch_bounds__(256, 4)
void ntt_pass2w(void *__restrict__ dst_, const u64 *__restrict__ scratch, const u64 *__restrict__ T2,
                long dst_stride, int nbatch, int nstore,
                const u32 *__restrict__ primes, const u64 *__restrict__ pinv, int prime0, int np_mod,
                const u32 *__restrict__ aux, long aux_stride, FoldGeom fg, const u64 *__restrict__ xtab) {
    // xtab: kOutU64Mul -- the table rows u64[row][L] the outputs are multiplied by; kOutModPNc -- psi^-j, u64[L]
    // np_mod > 0: the rows are several ciphertexts' worth of the same np_mod primes (batched operations); row r of the
    // whole call belongs to prime r mod np_mod and prime0 carries the row offset of this launch
    constexpr int L = 1 << LG, N1 = L / 64;
    constexpr bool INV = out_is_inverse(OUT);
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    int batch, tile;
    xcd_map(N1 / kP2wCols, batch, tile);
    if (batch >= nbatch) return;
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave index: uniform
    const int k1 = tile * kP2wCols + lane;
    const u64 *in = scratch + (long)batch * L + k1;
    const u64 *tw = T2 + k1;
    {
        u64 x[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const int j2 = 4 * a + r;
            u64 v = __builtin_nontemporal_load(&in[j2 * N1]);
            if (INV || j2 != 0) v = mulp(v, tw[j2 * N1]);
            x[a] = v;
        }
        dft_regs<16, false>(x);
        if (r == 0) StepAWrite<0, 0>::run(x, lds, lane);
        else if (r == 1) StepAWrite<1, 0>::run(x, lds, lane);
        else if (r == 2) StepAWrite<2, 0>::run(x, lds, lane);
        else StepAWrite<3, 0>::run(x, lds, lane);
    }
    __syncthreads();
    const int w = r;
    const P2Store A = pass2_store_args<LG, OUT>(dst_, dst_stride, nstore, primes, pinv, prime0, np_mod, aux, aux_stride, fg, xtab, batch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = 4 * i + w;
        u64 y[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) y[rr] = lds[(b * 4 + rr) * kP2wCols + lane];
        dft_regs<4, false>(y);                                           // y[bitrev4(c)] = X[b + 16c]
        pass2_store<LG, OUT>(y, b, k1, batch, A);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 1, wave-split form (ntt_pass1w): the N1-point column DFT is done as RA x 64 (RA = N1/64 = 4/8/16) and the
// 64-point stage again as 16 x 4 over wave-uniform quarters, so that every thread of the 512-thread workgroup
// carries 16 values in each of the three register stages (instead of 32 in two):
//   A   : item (col, b), b < 64: RA-point DFT over a (samples x[64*(64a + b) + j2], zero-padded half folded in),
//         times the inner twiddle w_N1^(b*c) (the only general multiplication of this pass), -> LDS [col][c][b]
//   B-A : thread (r, col, c): 16-point DFT over a' of the b = 4a' + r, times 2^(3*r*b'') (r is wave-uniform: one
//         specialised code path per wave, compile-time shifts), -> LDS [col][c][b''][r]   (same buffer, re-used)
//   B-B : thread (w, col, c): for b'' = 4i + w a 4-point DFT over r -> k1 = c + RA*(b'' + 16c''), stored to the slab.
// ---------------------------------------------------------------------------------------------------------------
static constexpr int kP1wThreads = 512;
template <int LG>
struct P1wGeom {
    static constexpr int N1 = (1 << LG) / 64, RA = N1 / 64, NC = 128 / RA;       // 128 (col, c) pairs per workgroup
    static constexpr int RS = 65;                         // row stride (u64) of both exchange layouts
    static constexpr int CS = RA * RS + 2;                // column stride, == 2 (mod 16) u64: conflict-free column lanes
    static constexpr int XCH = NC * CS;
    static constexpr int T1N = N1;
    static constexpr size_t bytes = (size_t)(XCH + T1N) * sizeof(u64);
};

template <int R, int B>
struct StepBWrite {
    static __device__ __forceinline__ void run(const u64 (&x)[16], u64 *row) {
        row[B * 4 + R] = TwShift<R, B>::run(x[bitrev<16>(B)]);
        if constexpr (B + 1 < 16) StepBWrite<R, B + 1>::run(x, row);
    }
};

template <int LG, int MODE>
__global__ __launch_bounds__(kP1wThreads, 2)
void ntt_pass1w(const void *__restrict__ src_, u64 *__restrict__ scratch,
                const u64 *__restrict__ T1, long src_stride, int nbatch, WindowArgs wa, const u64 *__restrict__ tw) {
    using G = P1wGeom<LG>;
    constexpr int L = 1 << LG, N1 = G::N1, RA = G::RA, NC = G::NC, T = kP1wThreads;
    constexpr int IA = NC * 64 / T;                       // stage-A items per thread (1 / 2 / 4), RA values each
    constexpr bool EXT = src_is_ext(MODE);
    constexpr int NA = EXT ? RA / 2 : RA;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 *xch = lds;
    u64 *t1 = lds + G::XCH;                               // t1[c*64 + b] = w_N1^(b*c)

    int batch, tile;
    xcd_map(64 / NC, batch, tile);
    if (batch >= nbatch) return;
    const int t = threadIdx.x;
    const int col0 = tile * NC;
    for (int i = t; i < G::T1N; i += T) t1[i] = T1[i];
    __syncthreads();

    // ---- stage A
#pragma unroll
    for (int it = 0; it < IA; ++it) {
        const int e = t + T * it;
        const int col = e % NC, b = e / NC;
        u64 x[RA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int idx = (a * 64 + b) * 64 + col0 + col;
            x[a] = load_sample<LG, MODE>(src_, src_stride, batch, idx, wa, tw);
        }
        dft_regs<RA, EXT>(x);
#pragma unroll
        for (int c = 0; c < RA; ++c) {
            u64 v = x[bitrev<RA>(c)];
            if (c != 0) v = mulp(v, t1[c * 64 + b]);
            xch[col * G::CS + c * G::RS + b] = v;
        }
    }
    __syncthreads();

    // ---- stage B-A: (r, col, c) with r wave-uniform
    const int r = __builtin_amdgcn_readfirstlane(t >> 7);
    const int u = t & 127;
    const int c = u % RA, col = u / RA;
    u64 *row = xch + col * G::CS + c * G::RS;
    u64 y[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) y[a] = row[4 * a + r];
    dft_regs<16, false>(y);
    __syncthreads();                                      // every read of the first layout done before it is overwritten
    if (r == 0) StepBWrite<0, 0>::run(y, row);
    else if (r == 1) StepBWrite<1, 0>::run(y, row);
    else if (r == 2) StepBWrite<2, 0>::run(y, row);
    else StepBWrite<3, 0>::run(y, row);
    __syncthreads();

    // ---- stage B-B
    u64 *out = scratch + (long)batch * L + (long)(col0 + col) * N1 + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int bb = 4 * i + r;
        u64 z[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) z[rr] = row[bb * 4 + rr];
        dft_regs<4, false>(z);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) out[RA * (bb + 16 * cc)] = z[bitrev<4>(cc)];
    }
}

// ===============================================================================================================
// Low-latency forms of both passes, for calls with FEW rows (a lone ciphertext operation transforms 1-100 rows).
// The kernels above give every thread 16 values in each of three dependent register stages: with a few dozen
// workgroups on a 256-CU chip their duration (13 + 9 us) is the latency of ONE workgroup, not a throughput.  Here a
// thread carries FOUR values per stage and every stage is a radix-4 decimation-in-frequency step done in place in
// LDS (general twiddles from a table in pass 1, powers of two in pass 2): 4x the threads per transform, a quarter of the
// work per thread between barriers.  More instructions per point than the 16-value forms (the in-register stages of
// those need no twiddle multiplications), so the host uses these only below a row count (cuhe_hip_set_ll_rows).
// Same inputs (every MODE), same slab layout scratch[j2][k1], same store epilogues (every OUT): bit-identical results.
// ===============================================================================================================
// x * 2^(3e) mod P for a run-time e in [0, 64): 2^k is 1 << k below 2^64, eps << (k - 64) up to 2^96, and 2^96 = -1
__device__ __forceinline__ u64 mul_w64(u64 x, int e) {
    int k = 3 * e;
    const bool neg = k >= 96;
    if (neg) k -= 96;
    const u64 c = k < 64 ? (1ULL << k) : ((u64)0xffffffffu << (k - 64));
    const u64 y = mulp(x, c);
    return neg ? negp(y) : y;
}
// in-place radix-4 DIF step on the four values x[a] = v[pos + a * Q] of a block (Q = a quarter of the block length):
// y[b] = sum_a x[a] w4^(ab), left in x[bitrev4(b)]
__device__ __forceinline__ void ll_dft4(u64 (&x)[4]) { dft_regs<4, false>(x); }

template <int LG> struct P1llGeom {
    static constexpr int N1 = (1 << LG) / 64, CW = 4, Q = N1 / 4, T = CW * Q, S = N1 + 8;      // T = N1 threads; S = 8 (mod 32): conflict-free columns
    static constexpr size_t bytes = (size_t)CW * S * sizeof(u64);
    static constexpr bool HALF = (LG == 15);              // N1 = 512 = 2 * 4^4: one radix-2 step first
};
// one radix-4 stage on block length M of the column buffer: thread i owns butterfly (block i / (M/4), offset i % (M/4))
template <int N1, int M>
__device__ __forceinline__ void ll_stage(u64 *buf, int i, const u64 *__restrict__ Wn1) {
    constexpr int Qm = M / 4;
    const int blk = i / Qm, o = i % Qm;
    u64 *v = buf + blk * M + o;
    u64 x[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) x[a] = v[a * Qm];
    ll_dft4(x);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        u64 y = x[bitrev<4>(b)];
        if constexpr (M > 4) { if (b) y = mulp(y, Wn1[(o * b) * (N1 / M)]); }
        v[b * Qm] = y;
    }
}
template <int N1> __host__ __device__ constexpr int ll_rev(int i, int digits4, bool lead2) {
    // position -> output index of the in-place DIF: base-4 digit reversal (after an optional leading binary digit)
    (void)lead2;
    int r = 0;
    for (int d = 0; d < digits4; ++d) { r = r * 4 + (i & 3); i >>= 2; }
    return r;
}

template <int LG, int MODE>
__global__ __launch_bounds__(P1llGeom<LG>::T)
void ntt_pass1_ll(const void *__restrict__ src_, u64 *__restrict__ scratch, const u64 *__restrict__ Wn1,
                  long src_stride, int nbatch, WindowArgs wa, const u64 *__restrict__ tw) {
    using G = P1llGeom<LG>;
    constexpr int L = 1 << LG, N1 = G::N1, CW = G::CW, Q = G::Q;
    constexpr bool EXT = src_is_ext(MODE);
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    int batch, tile;
    xcd_map(64 / CW, batch, tile);
    if (batch >= nbatch) return;
    const int col = threadIdx.x % CW, i = threadIdx.x / CW;            // adjacent lanes: adjacent columns (16 B of a u32 row)
    const int j2 = tile * CW + col;
    u64 *buf = lds + col * G::S;
    // ---- first step, from global memory: the samples j1 = i + Q a of column j2 (the zero-padded half is never read)
    u64 x[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (EXT && a >= 2) x[a] = 0;
        else x[a] = load_sample<LG, MODE>(src_, src_stride, batch, (i + Q * a) * 64 + j2, wa, tw);
    }
    if constexpr (G::HALF) {
        // N1 = 512: radix-2 over (a, a + 2), twiddle w_512^(i + Q a') on the differences; then blocks of 256
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const u64 u = x[a], v = x[a + 2];
            x[a] = EXT ? u : addp(u, v);
            const u64 d = EXT ? u : subp(u, v);
            buf[i + Q * (a + 2)] = mulp(d, Wn1[i + Q * a]);
            buf[i + Q * a] = x[a];
        }
        __syncthreads();
        ll_stage<N1, 256>(buf, i, Wn1); __syncthreads();
    } else {
        ll_dft4(x);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            u64 y = x[bitrev<4>(b)];
            if (b) y = mulp(y, Wn1[i * b]);
            buf[i + Q * b] = y;
        }
        __syncthreads();
        if constexpr (N1 == 1024) { ll_stage<N1, 256>(buf, i, Wn1); __syncthreads(); }
    }
    ll_stage<N1, 64>(buf, i, Wn1); __syncthreads();
    ll_stage<N1, 16>(buf, i, Wn1); __syncthreads();
    // ---- last step (blocks of 4) straight to the slab: position 4 i + b holds X[k1], k1 = digit reversal of the position
    {
        u64 *v = buf + 4 * i;
        u64 y[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) y[a] = v[a];
        ll_dft4(y);
        u64 *out = scratch + (long)batch * L + (long)j2 * N1;
        constexpr int D4 = (LG == 14) ? 4 : (LG == 15 ? 4 : 5);         // base-4 digits of a position (after the leading bit at 512)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int pos = 4 * i + b;
            int k1;
            if constexpr (G::HALF) {
                // pos = h * 256 + r, r in base 4 (4 digits): k1 = h + 2 * rev4(r)
                k1 = (pos >> 8) + 2 * ll_rev<N1>(pos & 255, 4, false);
            } else k1 = ll_rev<N1>(pos, D4, false);
            out[k1] = y[bitrev<4>(b)];
        }
    }
}

// pass 2: 16 adjacent k1 per workgroup (128-byte runs of the slab and of the output), 16 threads x 4 values per column
static constexpr int kP2llCols = 16, kP2llRS = 65;                    // odd row stride (u64): the 16 columns of a lane group fall on distinct bank pairs
template <int LG, int OUT>
__global__ __launch_bounds__(256)
void ntt_pass2_ll(void *__restrict__ dst_, const u64 *__restrict__ scratch, const u64 *__restrict__ T2,
                  long dst_stride, int nbatch, int nstore,
                  const u32 *__restrict__ primes, const u64 *__restrict__ pinv, int prime0, int np_mod,
                  const u32 *__restrict__ aux, long aux_stride, FoldGeom fg, const u64 *__restrict__ xtab) {
    constexpr int L = 1 << LG, N1 = L / 64;
    constexpr bool INV = out_is_inverse(OUT);
    __shared__ __attribute__((aligned(16))) u64 lds[kP2llCols * kP2llRS];
    int batch, tile;
    xcd_map(N1 / kP2llCols, batch, tile);
    if (batch >= nbatch) return;
    const int c = threadIdx.x & 15, t = threadIdx.x >> 4;               // column within the tile, butterfly index 0..15
    const int k1 = tile * kP2llCols + c;
    const u64 *in = scratch + (long)batch * L + k1;
    const u64 *tw = T2 + k1;
    u64 *v = lds + c * kP2llRS;
    u64 x[4];
    // block of 64: samples j2 = t + 16 a (outer twiddle on load), then * w_64^(t b)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j2 = t + 16 * a;
        u64 s = __builtin_nontemporal_load(&in[(long)j2 * N1]);
        if (INV || j2 != 0) s = mulp(s, tw[(long)j2 * N1]);
        x[a] = s;
    }
    ll_dft4(x);
#pragma unroll
    for (int b = 0; b < 4; ++b) v[t + 16 * b] = b ? mul_w64(x[bitrev<4>(b)], t * b) : x[0];
    __syncthreads();
    // blocks of 16: w_16^(o b) = w_64^(4 o b)
    {
        const int blk = t >> 2, o = t & 3;
        u64 *q = v + 16 * blk + o;
#pragma unroll
        for (int a = 0; a < 4; ++a) x[a] = q[4 * a];
        ll_dft4(x);
#pragma unroll
        for (int b = 0; b < 4; ++b) q[4 * b] = b ? mul_w64(x[bitrev<4>(b)], 4 * o * b) : x[0];
    }
    __syncthreads();
    // blocks of 4: position 4 t + b holds X[k2], k2 = rev16(t) + 16 b -- exactly what the store epilogue takes
#pragma unroll
    for (int a = 0; a < 4; ++a) x[a] = v[4 * t + a];
    ll_dft4(x);
    const P2Store A = pass2_store_args<LG, OUT>(dst_, dst_stride, nstore, primes, pinv, prime0, np_mod, aux, aux_stride, fg, xtab, batch);
    pass2_store<LG, OUT>(x, ((t & 3) << 2) | (t >> 2), k1, batch, A);
}

}  // namespace cuhe
