// icrt_mfma.cuh -- the inverse CRT with its column sums on the matrix cores (gfx950).
//
// cuhe/Base.cu:884-960 of the reference rebuilds a coefficient from its residues with one thread per coefficient and a
// 104-word register array; k_icrt (ops_kernels.cuh) turns that into  S = sum_i t_i (M / p_i),  t_i = x_i b_i mod p_i,
// minus q M, with np W multiply-adds (v_mad_u64_u32) per coefficient -- 1728 at 48 primes of 24 bits, 4681 vector lane
// instructions per coefficient with the residue products, the LDS reads and the fix-up (SQ_INSTS_VALU): issue bound
// (0.22 ms per 32 ciphertexts of x^32768 + 1, 1 TB/s of traffic).
//
// The sum is a matrix product: coefficients are columns, the contraction runs over (prime, digit of t_i) and the rows are
// the BYTES of the result.  With t_i in four base-128 digits (non-negative int8; primes below 2^28) and the constants
//     C_(i,a) = 128^a (M / p_i)          in signed base-256 digits c_(i,a)[d]
// the digit sums                 out[d] = sum_(i,a) digit_a(t_i) * c_(i,a)[d]                     (|out[d]| < 2^23)
// are exact in the int32 accumulators of v_mfma_i32_32x32x32_i8 (32 result bytes x 32 coefficients x 8 primes per
// instruction), and  S = sum_d out[d] 256^d.  What is left for the vector ALU is the residue products (leaner: see
// IcrtPrimeConst), four shift-adds per result word, the carry ripple and one conditional subtraction of M: 2555 vector
// lane instructions per coefficient (SQ_INSTS_VALU; 1.83x fewer), and the kernel runs 1.83x faster -- at the speed of
// its bytes.
//
// Lane map (one wave = one tile of 32 coefficients): lane = 32 h + c.  Second operand: lane (c, h) holds the digits of
// the primes 8 s + 4 h + e (e < 4) of coefficient c in K step s -- it formed those residue products itself, so the
// operand never passes through LDS.  First operand (the constants, staged in LDS once per workgroup): row rho of tile m
// is byte rho & 3 of result word  (rho >> 2 & 1) * WH + 4 m + (rho >> 3),  WH = 4 TILES: the result layout of the
// instruction (row = (reg & 3) + 8 (reg >> 2) + 4 h) then leaves lane (c, h) with the WH CONSECUTIVE words
// h WH .. h WH + WH - 1 of coefficient c, bytes of a word in four consecutive registers.  The ripple runs inside a lane;
// one cross-lane step (h = 0 -> h = 1) joins the halves.
//
// q = floor(sum_i t_i / p_i - 2^-30) clamped at 0 (f64: the sum is good to 2^-42) is the true quotient or one less, so
// r = S - q M lies in [0, 2 M) and ONE conditional subtraction makes the result exact.  -M is carried as the NW-word two's
// complement NM = 2^(32 NW) - M (NW = 2 WH >= W + 1): both S + q NM and r + NM are plain additions, and the carry out
// of r + NM is the comparison r >= M.
#pragma once
#include "ops_kernels.cuh"

namespace cuhe {

typedef int v16i __attribute__((ext_vector_type(16)));
// per-prime constants, 32 bytes (two 16-byte LDS reads): b_i = (M / p_i)^-1 mod p_i with its Shoup quotient
// bq = floor(b 2^32 / p), 1 / p_i.  x b mod p for ANY x < 2^32:  r = x b - floor(x bq / 2^32) p  (mod 2^32) lies in [0, 2 p)
// -- one high and two low 32-bit multiplies and a min, against the 64-bit Barrett form of mod_small (seven multiplies).
struct IcrtPrimeConst { u32 p, b, bq, pad0; double rp; u64 pad1; };
struct IcrtMfmaTab {
    const unsigned char *dig;       // [TILES][ksteps][64 lanes][16]: first operand, as the instruction reads it
    const IcrtPrimeConst *pc;       // [8 ksteps]: primes np .. 8 ksteps - 1 have b = 0 (residue product 0)
    const u32 *nm;                  // [8 TILES]: words of 2^(32 NW) - M
    int tiles, ksteps;
};
static constexpr int kIcrtMfmaThreads = 256, kIcrtMfmaTile = 32;
static inline size_t icrt_mfma_lds_bytes(int tiles, int ksteps) {
    return (size_t)tiles * ksteps * 1024 + (size_t)ksteps * 8 * sizeof(IcrtPrimeConst) + (size_t)8 * tiles * 4 +
           (size_t)(kIcrtMfmaThreads / 64) * 8 * tiles * (kIcrtMfmaTile + 1) * 4;
}
// base-128 digits of t < 2^28, one per byte
__device__ __forceinline__ int digits128(u32 t) {
    return (int)((t & 0x7fu) | ((t << 1) & 0x7f00u) | ((t << 2) & 0x7f0000u) | ((t << 3) & 0x7f000000u));
}

#ifndef CUHE_ICRT_WAVES
#define CUHE_ICRT_WAVES 3
#endif
template <int TILES>
__global__ __launch_bounds__(kIcrtMfmaThreads, CUHE_ICRT_WAVES)
void k_icrt_mfma(u32 *__restrict__ dst, const u32 *__restrict__ src, IcrtMfmaTab T, int np, int W, int mlen, int clen,
                 long src_ct_stride, long dst_ct_stride, int ncts, IcrtWindows wo) {
    constexpr int WH = 4 * TILES, NW = 8 * TILES, CB = kIcrtMfmaTile, RS = kIcrtMfmaTile + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char shraw[];
    const int ks = T.ksteps;
    v4i *tab = reinterpret_cast<v4i *>(shraw);                                        // [TILES * ks][64]
    IcrtPrimeConst *pc = reinterpret_cast<IcrtPrimeConst *>(tab + TILES * ks * 64);   // [8 ks]
    u32 *nm = reinterpret_cast<u32 *>(pc + ks * 8);                                   // [NW]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    u32 *wl = nm + NW + wave * NW * RS;                                               // [NW][33]: result words of this wave's tile (rows padded: the slab stores read down a column)
    for (int e = threadIdx.x; e < TILES * ks * 64; e += kIcrtMfmaThreads) tab[e] = reinterpret_cast<const v4i *>(T.dig)[e];
    for (int e = threadIdx.x; e < ks * 8 * 4; e += kIcrtMfmaThreads) reinterpret_cast<u64 *>(pc)[e] = reinterpret_cast<const u64 *>(T.pc)[e];
    for (int e = threadIdx.x; e < NW; e += kIcrtMfmaThreads) nm[e] = T.nm[e];
    __syncthreads();
    const int tiles_ct = (mlen + CB - 1) / CB, total = tiles_ct * ncts, stride = (int)gridDim.x * (kIcrtMfmaThreads / 64);
    const uint4 *nmq = reinterpret_cast<const uint4 *>(nm + h * WH);
    // The residues arrive through a prefetch that runs TWO K steps ahead of the arithmetic and straight on into the next
    // tile of this wave: (tile, step) positions form one stream, so the loads of a tile's first steps are in flight while
    // the previous tile is rippled and stored.  Addresses: a wave-uniform row pointer (scalar registers, advanced by scalar
    // adds) plus ONE 32-bit lane offset that never changes -- no vector arithmetic per load.
    const unsigned xoff = (unsigned)(4 * h) * (unsigned)clen + (unsigned)c;
    struct Pos { int tile, st; const u32 *row; bool live; };
    auto enter = [&](Pos &p) {                          // p.tile changed: its source rows and whether this lane has a coefficient
        p.live = false; p.row = src;
        if (p.tile < total) {
            const int ct = p.tile / tiles_ct, base = (p.tile % tiles_ct) * CB;
            p.live = c < mlen - base;
            p.row = src + (long)ct * src_ct_stride + base;
        }
    };
    auto advance = [&](Pos &p) {
        p.row += (long)8 * clen;
        if (++p.st == ks) { p.st = 0; p.tile += stride; enter(p); }
    };
    auto fetch = [&](const Pos &p, u32 (&x)[4]) {
        const u32 *r = p.row;
#pragma unroll
        for (int e = 0; e < 4; ++e, r += clen) x[e] = (p.live && 8 * p.st + 4 * h + e < np) ? r[xoff] : 0u;     // (padding primes: b = 0)
    };
    Pos pf{(int)blockIdx.x * (kIcrtMfmaThreads / 64) + wave, 0, src, false};
    enter(pf);
    u32 x0[4], x1[4];
    fetch(pf, x0); advance(pf);
    fetch(pf, x1); advance(pf);
    for (int tile = (int)blockIdx.x * (kIcrtMfmaThreads / 64) + wave; tile < total; tile += stride) {
        const int ct = tile / tiles_ct;
        const long base = (long)(tile % tiles_ct) * CB;
        const int nvalid = (int)min((long)CB, (long)mlen - base);
        const bool live = c < nvalid;
        v16i acc[TILES];
#pragma unroll
        for (int m = 0; m < TILES; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0;
        double a = 0.0;
        for (int st = 0; st < ks; ++st) {
            u32 x2[4];
            fetch(pf, x2); advance(pf);
            v4i B;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const IcrtPrimeConst k = pc[8 * st + 4 * h + e];
                const u32 r = x0[e] * k.b - __umulhi(x0[e], k.bq) * k.p;        // (x mod p) b mod p = x b mod p, in [0, 2 p): p < 2^28
                const u32 v = min(r, r - k.p);
                a += (double)v * k.rp;
                B[e] = digits128(v);
            }
#pragma unroll
            for (int m = 0; m < TILES; ++m)
                acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(tab[(m * ks + st) * 64 + lane], B, acc[m], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x0[e] = x1[e]; x1[e] = x2[e]; }
        }
        a += __shfl_xor(a, 32);                                    // both halves: the same sum (the addition commutes)
        const u32 q = (u32)fmax(a - 0x1p-30, 0.0);                 // true quotient or one less
        // words h WH + jj, jj < WH:  S - q M  as  S + q NM  (mod 2^(32 NW)); signed carries, everything fits 64 bits
        u32 wd[WH];
        long long carry = 0;
#pragma unroll
        for (int m = 0; m < TILES; ++m) {
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) {
                const int jj = 4 * m + r2;
                const uint4 n4 = nmq[jj >> 2];
                const u32 nmw = (jj & 3) == 0 ? n4.x : (jj & 3) == 1 ? n4.y : (jj & 3) == 2 ? n4.z : n4.w;
                const int plo = acc[m][4 * r2] + (acc[m][4 * r2 + 1] << 8), phi = acc[m][4 * r2 + 2] + (acc[m][4 * r2 + 3] << 8);
                const long long part = (long long)plo + ((long long)phi << 16) + carry;
                const u64 col = (u64)q * nmw + (u64)part;
                wd[jj] = (u32)col;
                carry = (long long)col >> 32;
            }
        }
        {   // the carry out of the low half enters the high half
            const int other = __shfl((int)carry, c);
            long long cc = h ? (long long)other : 0ll;
#pragma unroll
            for (int jj = 0; jj < WH; ++jj) {
                const long long t = (long long)wd[jj] + cc;
                wd[jj] = (u32)t;
                cc = t >> 32;
            }
        }
        {   // r >= M  <=>  r + NM carries out of word NW - 1; then r + NM = r - M is the result
            u32 d[WH], cy = 0;
#pragma unroll
            for (int jj = 0; jj < WH; ++jj) {
                const uint4 n4 = nmq[jj >> 2];
                const u32 nmw = (jj & 3) == 0 ? n4.x : (jj & 3) == 1 ? n4.y : (jj & 3) == 2 ? n4.z : n4.w;
                const u64 t = (u64)wd[jj] + nmw + cy;
                d[jj] = (u32)t;
                cy = (u32)(t >> 32);
            }
            const u32 c0 = (u32)__shfl((int)cy, c);
            u32 cc = h ? c0 : 0u;
#pragma unroll
            for (int jj = 0; jj < WH; ++jj) {
                const u64 t = (u64)d[jj] + cc;
                d[jj] = (u32)t;
                cc = (u32)(t >> 32);
            }
            const u32 ge = (u32)__shfl((int)(cy | cc), c + 32);       // the high half knows
#pragma unroll
            for (int jj = 0; jj < WH; ++jj) wd[jj] = ge ? d[jj] : wd[jj];
        }
        const bool win_regs = wo.win && wo.w == 16;                // 16-bit windows (every parameter set of the reference's examples) never
        if (win_regs) {                                            // straddle a word: they leave straight from the registers
            u32 *wp = wo.win + (long)ct * wo.ct_stride + base;     // wave-uniform row pointer, one 32-bit lane offset
            const unsigned woff = (unsigned)(2 * h * WH) * (unsigned)wo.clen + (unsigned)c;
            const int lim = live ? wo.k - 2 * h * WH : 0;          // windows of this lane (tied to the tile: the compares stay inside the loop
#pragma unroll
            for (int jj = 0; jj < WH; ++jj) {
                if (2 * jj < lim) wp[woff] = wd[jj] & 0xffffu;
                wp += wo.clen;
                asm volatile("" : "+s"(wp));                       // one running pointer: not 2 WH of them formed ahead in scalar registers
                if (2 * jj + 1 < lim) wp[woff] = wd[jj] >> 16;
                wp += wo.clen;
                asm volatile("" : "+s"(wp));
            }
        }
        if (!dst && (win_regs || !wo.win)) continue;
#pragma unroll
        for (int jj = 0; jj < WH; ++jj) wl[(h * WH + jj) * RS + c] = wd[jj];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (wo.win && !win_regs && live) {                       // win[j][coefficient] = bits [w j, w j + w) (cuhe/Base.cu:361-371); a half-wave: windows h, h + 2, ...
            u32 *wrow = wo.win + (long)ct * wo.ct_stride + base + c;
            const u32 mask = (u32)((1u << wo.w) - 1u);
            for (int j = h; j < wo.k; j += 2) {
                const int bit = wo.w * j, wi = bit >> 5;             // wi + 1 <= W < NW: the word above is always there (zero above the value)
                const u64 sv = (u64)wl[wi * RS + c] | (u64)wl[(wi + 1) * RS + c] << 32;
                wrow[(long)j * wo.clen] = (u32)(sv >> (bit & 31)) & mask;
            }
        }
        if (dst) {                                  // the slab [coefficient][W] leaves coalesced
            u32 *o = dst + (long)ct * dst_ct_stride + base * W;
            const int slab = nvalid * W, dc = 64 / W, dk = 64 % W;
            int c2 = lane / W, k = lane % W;
            for (int e = lane; e < slab; e += 64) {
                o[e] = wl[k * RS + c2];
                c2 += dc; k += dk;
                if (k >= W) { k -= W; ++c2; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace cuhe
