// ops_kernels.cuh -- CRT / ICRT / pointwise / polynomial-Barrett / modswitch /
// relinearisation kernels for gfx950.  Each kernel cites the reference kernel
// (cuhe/Base.cu) whose results it reproduces bit-for-bit; none of them is a
// translation: launches are batched over all CRT primes, constants live in
// HBM/L2 tables instead of 64 KB __constant__/textures, `%` is replaced by
// reciprocal multiplication and multiword carry chains use v_mad_u64_u32.
#pragma once
#include "modp.cuh"

namespace cuhe {

// per-prime constants resident on the device
struct PrimeTab {
    const u32 *p;        // [np]           CRT primes                      (const_p,  Base.cu:139)
    const u64 *pinv;     // [np]           floor(2^64 / p_i)
    const u32 *e64;      // [np]           2^64 mod p_i
    const u32 *pow32;    // [np][maxW]     2^(32k) mod p_i
    int maxW;
};

// ---------------------------------------------------------------- pointwise mod P
// z = x (op) y over np*L contiguous u64.  (ntt_mul / ntt_add: Base.cu:1036-1053;
// barrett_mul_un / barrett_mul_mn: Base.cu:927-949 are the same op with a table)
template <bool MUL>
__global__ __launch_bounds__(256)
void k_ntt_binop(u64 *__restrict__ z, const u64 *__restrict__ x, const u64 *__restrict__ y, long n2) {
    // n2 = number of u64 pairs
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        ulonglong2 a = reinterpret_cast<const ulonglong2 *>(x)[i];
        ulonglong2 b = reinterpret_cast<const ulonglong2 *>(y)[i];
        ulonglong2 r;
        if (MUL) { r.x = mulp(a.x, b.x); r.y = mulp(a.y, b.y); }
        else     { r.x = addp(a.x, b.x); r.y = addp(a.y, b.y); }
        reinterpret_cast<ulonglong2 *>(z)[i] = r;
    }
}
// second operand is ONE polynomial broadcast to every prime (ntt_mul_nx1 / ntt_add_nx1: Base.cu:1054-1075)
template <bool MUL>
__global__ __launch_bounds__(256)
void k_ntt_binop_nx1(u64 *__restrict__ z, const u64 *__restrict__ x, const u64 *__restrict__ s, int np, int L2) {
    // L2 = L/2 pairs per polynomial; grid.y = prime
    const int crt = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L2; i += gridDim.x * blockDim.x) {
        ulonglong2 a = reinterpret_cast<const ulonglong2 *>(x)[(long)crt * L2 + i];
        ulonglong2 b = reinterpret_cast<const ulonglong2 *>(s)[i];
        ulonglong2 r;
        if (MUL) { r.x = mulp(a.x, b.x); r.y = mulp(a.y, b.y); }
        else     { r.x = addp(a.x, b.x); r.y = addp(a.y, b.y); }
        reinterpret_cast<ulonglong2 *>(z)[(long)crt * L2 + i] = r;
    }
}

// ---------------------------------------------------------------- batched gate helpers (circuit evaluation on arrays)
// dst[t] = src[ia[t]] (*) src[ib[t]] for t < npairs: ciphertexts are `rows` rows of L (NTT domain); blockIdx.y = t
static __global__ __launch_bounds__(256)
void k_ntt_mul_pairs(u64 *__restrict__ dst, const u64 *__restrict__ src, const int *__restrict__ ia, const int *__restrict__ ib, long ct_pairs) {
    const int t = blockIdx.y;
    const ulonglong2 *x = reinterpret_cast<const ulonglong2 *>(src) + (long)ia[t] * ct_pairs;
    const ulonglong2 *y = reinterpret_cast<const ulonglong2 *>(src) + (long)ib[t] * ct_pairs;
    ulonglong2 *z = reinterpret_cast<ulonglong2 *>(dst) + (long)t * ct_pairs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ct_pairs; i += (long)gridDim.x * blockDim.x) {
        const ulonglong2 a = x[i], b = y[i];
        ulonglong2 r; r.x = mulp(a.x, b.x); r.y = mulp(a.y, b.y);
        z[i] = r;
    }
}
// dst[o] = sum of the CRT-domain ciphertexts srcs[list[t]] for t in [off[o], off[o+1])  (+ addc[o] on the constant
// coefficient), residues mod p_i.  A list entry e < nA addresses srcA[e], otherwise srcB[e - nA].  blockIdx.y = prime
// row, blockIdx.z = output.  This is a whole layer of cXor / cNot gates (CuHE.cu:122-215) in one launch.
template <int VEC>          // coefficients per thread: 4 (16-byte accesses; mlen, clen multiples of 4) or 1
__global__ __launch_bounds__(256)
void k_crt_combine(u32 *__restrict__ dst, const u32 *__restrict__ srcA, int nA, const u32 *__restrict__ srcB,
                   const int *__restrict__ off, const int *__restrict__ list, const int *__restrict__ addc,
                   PrimeTab pt, int np, int mlen, int clen) {
    const int o = blockIdx.z, i = blockIdx.y, idx = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (idx >= mlen) return;
    const u32 p = pt.p[i];
    u64 acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    for (int t = off[o]; t < off[o + 1]; ++t) {
        const int e = list[t];
        const u32 *s = e < nA ? srcA + ((long)e * np + i) * clen : srcB + ((long)(e - nA) * np + i) * clen;
        if (VEC == 4) { const uint4 x = *reinterpret_cast<const uint4 *>(s + idx); acc[0] += x.x; acc[1 % VEC] += x.y; acc[2 % VEC] += x.z; acc[3 % VEC] += x.w; }
        else acc[0] += s[idx];                           // residues < 2^32, a few dozen terms at most
    }
    if (idx == 0) acc[0] += (u32)addc[o];
    u32 r[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) r[v] = mod_small(acc[v], p, pt.pinv[i]);
    u32 *d = dst + ((long)o * np + i) * clen + idx;
    if (VEC == 4) *reinterpret_cast<uint4 *>(d) = make_uint4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    else d[0] = r[0];
}

// ---------------------------------------------------------------- CRT-domain ops
// inputs are residues < p_i, so (a+b)%p is one conditional subtract when a,b < p;
// the reference uses % (Base.cu:1088-1109) which also accepts unreduced inputs --
// we keep exact % semantics through mod_small.
template <int VEC>          // coefficients per thread: 4 (16-byte accesses) or 1
__global__ __launch_bounds__(256)
void k_crt_add(u32 *z, const u32 *a, const u32 *b, PrimeTab pt, int mlen, int clen) {       // (z may be a or b)
    const int crt = blockIdx.y, idx = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (idx >= mlen) return;
    const long o = (long)crt * clen + idx;
    const u32 p = pt.p[crt];
    const u64 m = pt.pinv[crt];
    if (VEC == 4) {
        const uint4 x = *reinterpret_cast<const uint4 *>(a + o), y = *reinterpret_cast<const uint4 *>(b + o);
        *reinterpret_cast<uint4 *>(z + o) = make_uint4(mod_small((u64)x.x + y.x, p, m), mod_small((u64)x.y + y.y, p, m),
                                                       mod_small((u64)x.z + y.z, p, m), mod_small((u64)x.w + y.w, p, m));
    } else z[o] = mod_small((u64)a[o] + b[o], p, m);
}
static __global__ __launch_bounds__(256)
void k_crt_add_nx1(u32 *__restrict__ z, const u32 *__restrict__ a, const u32 *__restrict__ s,
                   PrimeTab pt, int mlen, int clen) {
    const int crt = blockIdx.y, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= mlen) return;
    const long o = (long)crt * clen + idx;
    z[o] = mod_small((u64)a[o] + s[idx], pt.p[crt], pt.pinv[crt]);
}
// constant term only (Base.cu:1096-1100)
static __global__ void k_crt_add_int(u32 *__restrict__ z, const u32 *__restrict__ x, unsigned a,
                              PrimeTab pt, int np, int clen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const u32 p = pt.p[i];
    const u64 m = pt.pinv[i];
    z[(long)i * clen] = mod_small((u64)x[(long)i * clen] + mod_small(a, p, m), p, m);
}
// constant term times an integer (crt_mul_int, Base.cu:1078-1087)
static __global__ void k_crt_mul_int(u32 *__restrict__ z, const u32 *__restrict__ x, int a,
                              PrimeTab pt, int np, int clen) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    u64 t = (u64)x[(long)i * clen] * (u64)(long)a;        // same wrap-around as the reference's u64 *= int
    z[(long)i * clen] = mod_small(t, pt.p[i], pt.pinv[i]);
}

// ---------------------------------------------------------------- modulus switching (Base.cu:1112-1138)
// dst may alias src (row i only depends on rows i and np-1, and row np-1 is never written).
// A thread owns VEC consecutive coefficients (VEC = 4: 16-byte accesses, mlen and clen multiples of 4) and kModswPrimes
// target primes (blockIdx.y = group of primes): the dropped prime's residue and its parity fix are formed once per
// coefficient, not once per (coefficient, prime); no branch on the sign of the difference (a multiple of p is added first).
// Round 4: the one-element-per-thread form ran at 1 TB/s on arrays of 64 ciphertexts (profiles/r04_elementwise_ab.txt).
static constexpr int kModswPrimes = 4;
template <int VEC>
__device__ __forceinline__ void modswitch_body(u32 *__restrict__ dst, const u32 *__restrict__ src, const PrimeTab &pt,
                                               const u32 *__restrict__ invp, int np, int mlen, int clen, int modmsg) {
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (idx >= mlen) return;
    const u32 ptl = pt.p[np - 1];
    u32 raw[VEC];
    if (VEC == 4) { const uint4 v = *reinterpret_cast<const uint4 *>(src + (long)(np - 1) * clen + idx); raw[0] = v.x; raw[1 % VEC] = v.y; raw[2 % VEC] = v.z; raw[3 % VEC] = v.w; }
    else raw[0] = src[(long)(np - 1) * clen + idx];
    long long dirty[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const int d = (int)raw[v];
        const int ep = modmsg == 2 ? (d & 1) : d % modmsg;          // (the residue is below 2^31: non-negative as an int)
        dirty[v] = d;
        if (ep != 0) dirty[v] += (raw[v] > ((ptl - 1) / 2)) ? -(long long)ep * (long long)ptl : (long long)ep * (long long)ptl;
    }
    const int i1 = min((int)(blockIdx.y + 1) * kModswPrimes, np - 1);
    for (int i = blockIdx.y * kModswPrimes; i < i1; ++i) {
        const u32 p = pt.p[i];
        const u64 m = pt.pinv[i];
        const u32 inv = invp[(np - 1) * (np - 2) / 2 + i];
        const u64 lift = (u64)p << (11 + __clz(p));                  // a multiple of p in [2^42, 2^43): above every |src - dirty| (< 2^32 + modmsg p_t < 2^42: checked by cuhe_hip_init)
        u32 x[VEC], r[VEC];
        if (VEC == 4) { const uint4 v = *reinterpret_cast<const uint4 *>(src + (long)i * clen + idx); x[0] = v.x; x[1 % VEC] = v.y; x[2 % VEC] = v.z; x[3 % VEC] = v.w; }
        else x[0] = src[(long)i * clen + idx];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const u32 t = mod_small((u64)((long long)x[v] - dirty[v] + (long long)lift), p, m);       // (src - dirty) mod p
            r[v] = mod_small((u64)t * inv, p, m);                                                     // times p_t^-1 mod p
        }
        if (VEC == 4) *reinterpret_cast<uint4 *>(dst + (long)i * clen + idx) = make_uint4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
        else dst[(long)i * clen + idx] = r[0];
    }
}
template <int VEC>
__global__ __launch_bounds__(256)
void k_modswitch(u32 *__restrict__ dst, const u32 *__restrict__ src, PrimeTab pt,
                 const u32 *__restrict__ invp, int np, int mlen, int clen, int modmsg, long src_ct_stride, long dst_ct_stride) {
    // blockIdx.z: ciphertext of a batched call (the result has np-1 rows, so a packed result array has a smaller stride than its source)
    modswitch_body<VEC>(dst + (long)blockIdx.z * dst_ct_stride, src + (long)blockIdx.z * src_ct_stride, pt, invp, np, mlen, clen, modmsg);
}

// ---------------------------------------------------------------- polynomial Barrett pieces
// Last step of the NTT-based Barrett reduction, fused (replaces barrett_sub x2, barrett_sub_mc and the strided gather
// of Base.cu:951-1001):  f = polynomial to reduce (row stride nlen, residues < p), qrow = quotient q stored at offset
// mlen of its row, mq = ((m - x^n) q) mod p.  r = f - q x^n - (m - x^n) q has degree <= n; for idx < n the q x^n term
// does not contribute, and the reference's correction subtracts m once more when the coefficient of x^n is non-zero.
static __global__ __launch_bounds__(256)
void k_barrett_final(u32 *__restrict__ dst, const u32 *__restrict__ f, const u32 *__restrict__ qrow, const u32 *__restrict__ mq,
                     const u32 *__restrict__ m_crt, PrimeTab pt, int mlen, int clen, int nlen, int np_mod) {
    const int crt = blockIdx.y, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= clen) return;
    const long base = (long)crt * nlen;
    const int pi = np_mod > 0 ? crt % np_mod : crt;          // batched calls: row -> prime
    const u32 p = pt.p[pi];
    u32 r = 0;
    if (idx < mlen) {
        u32 a = f[base + idx], b = mq[base + idx];
        r = a >= b ? a - b : a + p - b;
        // coefficient mlen of r (wave-uniform per row): f[n] - q[0] - mq[n]
        u32 t = f[base + mlen], q0 = qrow[base + mlen], b2 = mq[base + mlen];
        t = t >= q0 ? t - q0 : t + p - q0;
        t = t >= b2 ? t - b2 : t + p - b2;
        if (t != 0 && idx < mlen - 1) {
            const u32 s = m_crt[(long)pi * clen + idx];
            r = r >= s ? r - s : r + p - s;
        }
    }
    dst[(long)crt * clen + idx] = r;
}

// ---- folded form of the generic reduction (cuhe_transforms.hip: barrett_impl; FoldGeom and fold_g live in ntt_kernels.cuh)
// A[j] = g[D-1-j] for j < Kq (the top of g, reversed), zero up to Lh/2: input of the half-length forward transform
static __global__ __launch_bounds__(256)
void k_fold_top_rev(u32 *__restrict__ A, const u32 *__restrict__ f, PrimeTab pt, FoldGeom G, int nlen, int np_mod) {
    const int crt = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= G.Lh / 2) return;
    const u32 p = pt.p[np_mod > 0 ? crt % np_mod : crt];
    A[(long)crt * (G.Lh / 2) + j] = j < G.Kq ? fold_g(f + (long)crt * nlen, G.D - 1 - j, G, p) : 0u;
}
// (the quotient reversal and the final subtraction are store epilogues of the inverse transforms: kOutModPRevQ, kOutFoldFinal)

// fast exact reduction when the modulus is x^n + 1 (m = 2n a power of two):
//   r[i] = f[i] - f[i+n]           (f has degree <= 2n-2)
// and when m is prime (Phi_m = 1 + x + ... + x^(m-1), n = m-1):
//   fold mod x^m - 1, then subtract the coefficient of x^(m-1) from every term.
// Both give the same residues as the NTT-based Barrett of Operations.cu:460-501.
template <int KIND>   // 0: x^n+1 ; 1: prime m
__global__ __launch_bounds__(256)
void k_reduce_special(u32 *__restrict__ dst, const u32 *__restrict__ f, PrimeTab pt, int n, int clen, int nlen, int np_mod) {
    const int crt = blockIdx.y, idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= clen) return;
    const u32 *row = f + (long)crt * nlen;
    const u32 p = pt.p[np_mod > 0 ? crt % np_mod : crt];
    u32 r = 0;
    if (idx < n) {
        if (KIND == 0) {
            u32 a = row[idx], b = row[idx + n];
            r = a >= b ? a - b : a + p - b;
        } else {
            const int m = n + 1;
            // folded g[i] = f[i] + f[i+m]  (i+m <= 2n-2 = 2m-4  =>  i <= m-4)
            u32 a = row[idx];
            if (idx + m <= 2 * n - 2) { a += row[idx + m]; if (a >= p) a -= p; }
            u32 top = row[m - 1];       // g[m-1] = f[m-1]  (f[2m-1] is beyond the degree bound)
            r = a >= top ? a - top : a + p - top;
        }
    }
    dst[(long)crt * clen + idx] = r;
}

// ---------------------------------------------------------------- relinearisation inner product
// dst[i][idx] = sum_{j<k} c[j][idx] * ek[i][j][idx] mod P   (relinMulAddPerCrt, Base.cu:1024-1033,
// launched once for ALL primes; keys are device resident instead of streamed over PCIe per call,
// Relinearization.cu:80-87).  128-bit products are accumulated unreduced in 160 bits and folded once.
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
// 128-bit accumulate of a*b into (lo, hi, top)
__device__ __forceinline__ void mac192(u64 a, u64 b, u64 &lo, u64 &hi, u32 &top) {
    u64 pl = a * b, ph = __umul64hi(a, b);
    u64 nl = lo + pl;
    u64 cy = nl < lo;
    lo = nl;
    u64 nh = hi + ph;
    u32 c2 = nh < hi;
    nh += cy;
    c2 += (nh < cy);
    hi = nh;
    top += c2;
}
__device__ __forceinline__ u64 fold192(u64 lo, u64 hi, u32 top) {
    // value = lo + hi*2^64 + top*2^128 ; 2^128 = -2^32 (mod P)
    u64 r = reduce128(lo, hi);
    u64 t = (u64)top << 32;             // top < 2^8: canonical
    return subp(r, t);
}
// Two adjacent coefficients per thread (16 B/lane loads), PB primes and BB ciphertexts per block: every window value
// c[b][j][idx] fetched once serves PB key streams (the cache-resident operand costs 1/PB of the HBM key traffic), and
// in a batched call every key value fetched once serves BB ciphertexts (8*k*L key bytes per prime are the algorithmic
// bytes of ONE relinearisation; a batch of B shares them).  blockIdx.z = group of BB ciphertexts.
// blockIdx.x enumerates (column tile, prime block) pairs.  MAP 0: column tile fastest (a prime block's key rows are
// streamed in address order).  MAP 1, used by the batched call: workgroup ids go round-robin over the 8 XCDs, so
// id % 8 picks the XCD, and within an XCD the prime block varies fastest for a fixed column tile, which keeps the
// window tile (BB * k * 512 columns * 8 B) in that XCD's L2 for its np/PB consecutive prime blocks.
template <int PB, int BB, int MAP = 0>
__global__ __launch_bounds__(256)
void k_relin_mac(u64 *__restrict__ dst, const u64 *__restrict__ c, const u64 *__restrict__ ek,
                 int k, long ek_prime_stride, int L, int np, long c_ct_stride, long dst_ct_stride, int ncts) {
    const int b0 = blockIdx.z * BB;
    const int ny = (np + PB - 1) / PB;
    int xt, i0;
    if constexpr (MAP == 1) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;    // gridDim.x = (L/512) * ny, L/512 a multiple of 8
        xt = (slot / ny) * 8 + xcd;                                // column tile
        i0 = (slot % ny) * PB;
    } else {
        const int nx = gridDim.x / ny;
        xt = blockIdx.x % nx;                                      // column tile fastest, then prime block
        i0 = (blockIdx.x / nx) * PB;
    }
    const int idx2 = xt * blockDim.x + threadIdx.x;            // pair index; L/2 is a multiple of 256
    const long L2 = L / 2;
    const u64x2 *cc[BB];
#pragma unroll
    for (int b = 0; b < BB; ++b) cc[b] = reinterpret_cast<const u64x2 *>(c + (long)min(b0 + b, ncts - 1) * c_ct_stride) + idx2;
    const u64x2 *e[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int i = min(i0 + q, np - 1);                     // clamp: tail block recomputes the last prime
        e[q] = reinterpret_cast<const u64x2 *>(ek + (long)i * ek_prime_stride) + idx2;
    }
    u64 lo0[BB][PB], hi0[BB][PB], lo1[BB][PB], hi1[BB][PB]; u32 top0[BB][PB], top1[BB][PB];
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int q = 0; q < PB; ++q) { lo0[b][q] = hi0[b][q] = lo1[b][q] = hi1[b][q] = 0; top0[b][q] = top1[b][q] = 0; }
    for (int j = 0; j < k; ++j) {
        u64x2 a[BB];
#pragma unroll
        for (int b = 0; b < BB; ++b) a[b] = cc[b][(long)j * L2];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const u64x2 kv = __builtin_nontemporal_load(&e[q][(long)j * L2]);
#pragma unroll
            for (int b = 0; b < BB; ++b) {
                mac192(a[b].x, kv.x, lo0[b][q], hi0[b][q], top0[b][q]);
                mac192(a[b].y, kv.y, lo1[b][q], hi1[b][q], top1[b][q]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            if (i0 + q < np && b0 + b < ncts) {
                u64x2 r;
                r.x = fold192(lo0[b][q], hi0[b][q], top0[b][q]);
                r.y = fold192(lo1[b][q], hi1[b][q], top1[b][q]);
                reinterpret_cast<u64x2 *>(dst + (long)(b0 + b) * dst_ct_stride + (long)(i0 + q) * L)[idx2] = r;
            }
        }
}

// ---- batched form with the window tile resident in LDS.
// The blocking above keeps PB x BB accumulators per thread and re-reads every window value once per prime block; in a
// batch that traffic (L2 misses: window tiles + keys) and the multiply-add's instruction count bound the kernel
// together (profiles/r01_experiments_log.txt).  Here a workgroup owns 32 columns of BB ciphertexts: their k window
// rows (BB * k * 256 B) are loaded into LDS ONCE, then the 8 prime groups of the workgroup (thread = (column, group))
// walk over all primes, PB at a time: every key value is fetched from HBM exactly once per BB ciphertexts and every
// window value from global memory exactly once.  With the traffic gone the kernel is VALU-bound, so the multiply-add
// itself is restructured (AccSplit below): 8 VALU instructions instead of 24.
// Split accumulator: with a = a0 + a1 2^32, b = b0 + b1 2^32 the sum of products is
//     S = sum a0 b0  +  2^64 sum a1 b1  +  2^32 (sum a0 b1 + sum a1 b0),
// three independent 64-bit running sums with a carry counter each.  v_mad_u64_u32 already adds a 64-bit operand and
// reports the carry, so one multiply-add of the inner product is four v_mad_u64_u32 and four v_addc_co_u32 = 8 VALU
// instructions (the 128-bit product + 160-bit add costs 15 with a hand-written carry chain, 24 in C).  The three
// sums are recombined modulo P once per output.
struct AccSplit { u64 s00, s11, sx; u32 c00, c11, cx; };
// The four multiply-adds first, then the four carry adds: every carry (VCC and three SGPR pairs) is read three or more
// instructions after the VALU instruction that wrote it, which covers the two wait states gfx940/950 need between a VALU
// write of an SGPR / VCC and a VALU read of it (nothing pads the inside of an asm string; tools/asm_hazard_check.py).
__device__ __forceinline__ void mac_split(u64 a, u64 b, AccSplit &s) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 tA, tB, tC;
    asm("v_mad_u64_u32 %[s00], vcc, %[a0], %[b0], %[s00]\n\t"
        "v_mad_u64_u32 %[sx], %[tA], %[a0], %[b1], %[sx]\n\t"
        "v_mad_u64_u32 %[s11], %[tB], %[a1], %[b1], %[s11]\n\t"
        "v_mad_u64_u32 %[sx], %[tC], %[a1], %[b0], %[sx]\n\t"
        "v_addc_co_u32_e32 %[c00], vcc, 0, %[c00], vcc\n\t"
        "v_addc_co_u32_e64 %[cx], %[tA], 0, %[cx], %[tA]\n\t"
        "v_addc_co_u32_e64 %[c11], %[tB], 0, %[c11], %[tB]\n\t"
        "v_addc_co_u32_e64 %[cx], %[tC], 0, %[cx], %[tC]"
        : [s00] "+v"(s.s00), [s11] "+v"(s.s11), [sx] "+v"(s.sx), [c00] "+v"(s.c00), [c11] "+v"(s.c11), [cx] "+v"(s.cx),
          [tA] "=&s"(tA), [tB] "=&s"(tB), [tC] "=&s"(tC)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1)
        : "vcc");
}
// S mod P with 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32:
//   S = (s00 + c00 2^64) + 2^64 (s11 + c11 2^64) + 2^32 (sx + cx 2^64)
__device__ __forceinline__ u64 fold_split(const AccSplit &s) {
    u64 r = reduce128(s.s00, s.s11);                       // s00 + 2^64 s11
    r = addp(r, shlp<32>(canon(s.sx)));                    // + 2^32 sx
    r = addp(r, mulp_u32(kEps, s.c00));                    // + c00 2^64
    r = subp(r, (u64)s.c11 << 32);                         // + c11 2^128 = - c11 2^32   (c11 < 2^32: canonical)
    r = subp(r, (u64)s.cx);                                // + cx 2^96  = - cx
    return r;
}
static constexpr int kMacLdsThreads = 256;
template <int PB, int BB, int CB>
__global__ __launch_bounds__(kMacLdsThreads)
void k_relin_mac_lds(u64 *__restrict__ dst, const u64 *__restrict__ c, const u64 *__restrict__ ek,
                     int k, long ek_prime_stride, int L, int np, long c_ct_stride, long dst_ct_stride, int ncts) {
    extern __shared__ __attribute__((aligned(16))) u64 wl[];   // [BB][k][CB]
    constexpr int NG = kMacLdsThreads / CB;                  // prime groups per workgroup
    const int col = threadIdx.x % CB, pg = threadIdx.x / CB;
    // 1-D grid over (column tile, group of BB ciphertexts).  Workgroup ids go round-robin over the 8 XCDs (id % 8); within
    // an XCD consecutive ids take the ciphertext groups of ONE column tile, so that the groups that read the same key rows
    // run next to each other on one L2: a key value then comes from HBM once per batch instead of once per group
    // (L / CB is a multiple of 8).
    const int ngroups = (ncts + BB - 1) / BB;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long col0 = (long)((slot / ngroups) * 8 + xcd) * CB;
    const int b0 = (slot % ngroups) * BB;
    for (int e = threadIdx.x; e < BB * k * CB; e += CB * NG) {
        const int cc = e % CB, j = (e / CB) % k, b = e / (CB * k);
        wl[e] = c[(long)min(b0 + b, ncts - 1) * c_ct_stride + (long)j * L + col0 + cc];
    }
    __syncthreads();
    const u64 *mine = wl + col;
    for (int i0 = pg * PB; i0 < np; i0 += NG * PB) {
        AccSplit s[BB][PB];
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int q = 0; q < PB; ++q) s[b][q] = AccSplit{0, 0, 0, 0, 0, 0};
        const u64 *e[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) e[q] = ek + (long)min(i0 + q, np - 1) * ek_prime_stride + col0 + col;
        // software pipeline over blocks of JB windows: the key values of the NEXT block are requested before the
        // multiply-adds of the current one, so that JB * PB loads per thread are in flight (two waves per SIMD alone
        // keep too few bytes in flight to stream the keys at HBM speed: 3.4 TB/s measured with a distance of one)
        constexpr int JB = 4;
        u64 kv[JB][PB], kn[JB][PB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
            for (int q = 0; q < PB; ++q) kv[jj][q] = __builtin_nontemporal_load(&e[q][(long)min(jj, k - 1) * L]);
        for (int j0 = 0; j0 < k; j0 += JB) {
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int q = 0; q < PB; ++q) kn[jj][q] = __builtin_nontemporal_load(&e[q][(long)min(j0 + JB + jj, k - 1) * L]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                const int j = j0 + jj;
                if (j < k) {
#pragma unroll
                    for (int b = 0; b < BB; ++b) {
                        const u64 a = mine[(b * k + j) * CB];
#pragma unroll
                        for (int q = 0; q < PB; ++q) mac_split(a, kv[jj][q], s[b][q]);
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int q = 0; q < PB; ++q) kv[jj][q] = kn[jj][q];
        }
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int q = 0; q < PB; ++q)
                if (i0 + q < np && b0 + b < ncts)
                    dst[(long)(b0 + b) * dst_ct_stride + (long)(i0 + q) * L + col0 + col] = fold_split(s[b][q]);
    }
}

// ---------------------------------------------------------------- key-switch inner product on the matrix cores
// For one column c the inner product  out[b][i] = sum_j win[b][j] * ek[j][i]  (b: ciphertexts of a batch, j < k windows,
// i < np primes; cuhe/Relinearization.cu:76-88 evaluates it one ciphertext at a time) is a (B x k) x (k x np) matrix
// product over Z_P -- a real contraction over j, not a reshaped stream.  Written in signed base-256 digits
//     win = sum_la a_la 256^la,   ek = sum_lb e_lb 256^lb      (a_la, e_lb in [-128, 128), of a representative of the
//                                                               value in [-0x8080808080808080, 2^64 - 0x8080808080808080))
// it becomes 64 int8 products per 64-bit product, accumulated exactly in int32 by v_mfma_i32_16x16x64_i8:
//     D_t[b][i] = sum over la + lb = t of  sum_j a_la[b][j] * e_lb[j][i]            (15 accumulator tiles, |D_t| < 2^24)
//     out[b][i] = sum_t D_t 256^t  mod P                                           (a few dozen VALU operations per output)
// against 8 VALU instructions (four of them v_mad_u64_u32) per 64-bit product in k_relin_mac_lds.  One MFMA tile is
// 16 ciphertexts x 16 primes x 64 windows; the K order inside an instruction is irrelevant as long as both operands use
// the same one, so only the row / column maps of the instruction matter (rows = lane & 15 of the first operand, columns
// = lane & 15 of the second; result: column = lane & 15, row = 4 (lane >> 4) + register).
// The key digits are laid out once (k_ek_digits) exactly as the second operand wants them: for every column, 16-prime
// tile and digit, the 64-window steps as [lane][16 bytes] and the tail of K mod 64 windows as [lane group][prime][8 or 16
// bytes] holding only the groups that exist -- the kernel streams the keys from HBM once per 16 ciphertexts with
// wave-wide contiguous loads and no padding bytes.
typedef int v4i __attribute__((ext_vector_type(4)));
static constexpr u64 kDigC = 0x8080808080808080ull, kDigT = 0x7F7F7F7F7F7F7F80ull;
// bytes of the result = two's complement signed digits of a' = (a < T ? a : a - P), a' == a (mod P)
__device__ __forceinline__ u64 signed_digits(u64 a) {
    const u64 y = a + kDigC + (a >= kDigT ? 0xffffffffull : 0ull);       // a' + C  in [0, 2^64)   (-P == 2^32 - 1 mod 2^64)
    return y ^ kDigC;                                                   // unsigned byte u -> signed digit u - 128
}
struct MacDigGeom {
    int nfull;           // K / 64 steps of 64 windows
    int tail;            // 0, 32 (K mod 64 in 1..32: 8-byte lanes) or 64 (33..63: 16-byte lanes)
    int tail_groups;     // lane groups of the tail step that hold windows
    int lb_bytes;        // bytes of one (column, prime tile, digit) block
    int npt;             // prime tiles (of the top level)
};
static constexpr int kMacMfmaCols = 8, kMacMfmaCts = 16;
// key digits: one thread per (column, prime slot, chunk of 16 / 8 windows); reads coalesced over columns
static __global__ __launch_bounds__(256)
void k_ek_digits(unsigned char *__restrict__ ekd, const u64 *__restrict__ ek, int K, int np, int L, long ek_prime_stride, MacDigGeom G) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int slot = blockIdx.y, pt = slot >> 4, n = slot & 15, i = slot;
    const int q = blockIdx.z;                               // chunk: full steps first (4 groups each), then the tail groups
    if (c >= L) return;
    int j0, nj, off;
    if (q < G.nfull * 4) { j0 = q * 16; nj = 16; off = (q >> 2) * 1024 + ((q & 3) * 16 + n) * 16; }
    else {
        const int g = q - G.nfull * 4;
        if (G.tail == 32) { j0 = G.nfull * 64 + g * 8; nj = 8; off = G.nfull * 1024 + (g * 16 + n) * 8; }
        else { j0 = G.nfull * 64 + g * 16; nj = 16; off = G.nfull * 1024 + (g * 16 + n) * 16; }
    }
    u64 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = j0 + t;
        w[t] = (t < nj && j < K && i < np) ? signed_digits(ek[(long)i * ek_prime_stride + (long)j * L + c]) : 0ull;
    }
    unsigned char *base = ekd + ((long)c * G.npt + pt) * 8 * G.lb_bytes + off;
#pragma unroll
    for (int lb = 0; lb < 8; ++lb) {
        u32 d[4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
            d[x] = (u32)((w[4 * x] >> (8 * lb)) & 0xff) | (u32)((w[4 * x + 1] >> (8 * lb)) & 0xff) << 8 |
                   (u32)((w[4 * x + 2] >> (8 * lb)) & 0xff) << 16 | (u32)((w[4 * x + 3] >> (8 * lb)) & 0xff) << 24;
        u32 *o = (u32 *)(base + (long)lb * G.lb_bytes);
        o[0] = d[0]; o[1] = d[1];
        if (nj == 16) { o[2] = d[2]; o[3] = d[3]; }
    }
}
// sum_t D_t 256^t mod P with 2^64 = 2^32 - 1, 2^96 = -1 (|D_t| < 2^24, t < 15).  The accumulators start at kDigBias, so the
// fold sees POSITIVE 25-bit numbers E_t = D_t + 2^24 and needs no sign handling: four words  w_q = sum_r E_(4q+r) 256^r
// (three v_mad_u64_u32 each, < 2^50) with  sum_q w_q 2^(32q) = (w_0 - w_2 - w_3) + 2^32 (w_1 + w_2)  (mod P), and the bias
// 2^24 sum_t 256^t leaves through the constant that also keeps the 128-bit sum non-negative.
static constexpr int kDigBias = 1 << 24;
constexpr u64 dig_bias_mod_p() {
    unsigned __int128 s = 0, w = (unsigned __int128)kDigBias;
    for (int t = 0; t < 15; ++t) { s = (s + w) % kP; w = (w << 8) % kP; }
    return (u64)s;
}
__device__ __forceinline__ u64 fold_digits(const u32 (&E)[15]) {
    u64 w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w[q] = (u64)E[4 * q] + ((u64)E[4 * q + 1] << 8) + ((u64)E[4 * q + 2] << 16);
        if (4 * q + 3 < 15) w[q] += (u64)E[4 * q + 3] << 24;
    }
    constexpr unsigned __int128 kFix = 2 * (unsigned __int128)kP - dig_bias_mod_p();       // >= P > 2^51 >= w_2 + w_3
    const unsigned __int128 U = kFix + w[0] - w[2] - w[3] + ((unsigned __int128)(w[1] + w[2]) << 32);      // in [0, 2^84)
    return mad_eps((u32)(U >> 64), (u64)U);                      // lo + hi (2^32 - 1), canonical (hi < 2^20)
}
template <int NFULL, int TAIL> struct MacFrag {
    v4i f[8][NFULL ? NFULL : 1];
    v4i t64[TAIL == 64 ? 8 : 1];
    long t32[TAIL == 32 ? 8 : 1];
};
template <int NFULL, int TAIL>
__device__ __forceinline__ void mac_load_keys(MacFrag<NFULL, TAIL> &B, const unsigned char *p, int lb_bytes, int lane, int tail_groups) {
#pragma unroll
    for (int lb = 0; lb < 8; ++lb) {
        const unsigned char *q = p + (long)lb * lb_bytes;
#pragma unroll
        for (int s = 0; s < NFULL; ++s) B.f[lb][s] = *(const v4i *)(q + s * 1024 + lane * 16);
        if (TAIL == 32) B.t32[lb] = (lane >> 4) < tail_groups ? *(const long *)(q + NFULL * 1024 + lane * 8) : 0l;
        if (TAIL == 64) B.t64[lb] = (lane >> 4) < tail_groups ? *(const v4i *)(q + NFULL * 1024 + lane * 16) : v4i{0, 0, 0, 0};
    }
}
__device__ __forceinline__ u32 pack_digit(u64 w0, u64 w1, u64 w2, u64 w3, int la) {
    return (u32)((w0 >> (8 * la)) & 0xff) | (u32)((w1 >> (8 * la)) & 0xff) << 8 | (u32)((w2 >> (8 * la)) & 0xff) << 16 |
           (u32)((w3 >> (8 * la)) & 0xff) << 24;
}
// the eight digit words of four window words at once: o[la] = { byte la of w0, w1, w2, w3 } -- a 4 x 4 byte transpose of the low
// dwords (la < 4) and one of the high dwords, 8 v_perm_b32 each (2 per output word against 3-4 shift / mask / or operations
// per word for pack_digit as hipcc lowers it).  v_perm_b32 D, S0, S1, sel: byte k of D = byte sel_k of { S1 (0..3), S0 (4..7) }.
__device__ __forceinline__ void transpose_digits(u32 (&o)[8], u64 w0, u64 w1, u64 w2, u64 w3) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const u32 d0 = (u32)(w0 >> (32 * h)), d1 = (u32)(w1 >> (32 * h)), d2 = (u32)(w2 >> (32 * h)), d3 = (u32)(w3 >> (32 * h));
        const u32 a01 = __builtin_amdgcn_perm(d1, d0, 0x05010400u);     // d0.b0 d1.b0 d0.b1 d1.b1
        const u32 b01 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);     // d0.b2 d1.b2 d0.b3 d1.b3
        const u32 a23 = __builtin_amdgcn_perm(d3, d2, 0x05010400u);
        const u32 b23 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
        o[4 * h + 0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);    // a01.b0 a01.b1 a23.b0 a23.b1
        o[4 * h + 1] = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
        o[4 * h + 2] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
        o[4 * h + 3] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
    }
}
static constexpr int kMacMfmaThreads = 256;
template <int NFULL, int TAIL>
__global__ __launch_bounds__(kMacMfmaThreads, (2 * NFULL + (TAIL == 64 ? 2 : TAIL == 32 ? 1 : 0)) <= 3 ? 2 : 1)
void k_relin_mac_mfma(u64 *__restrict__ dst, const u64 *__restrict__ c, const unsigned char *__restrict__ ekd,
                      int k, int L, int np, long c_ct_stride, long dst_ct_stride, int ncts, MacDigGeom G) {
    extern __shared__ __attribute__((aligned(16))) u64 sa[];          // [8 columns][16 ciphertexts x JS + 1]: signed-digit words
    const int npt = (np + 15) >> 4, NPAD = npt * 16;
    const int JS = (k + 2) & ~1, CS = kMacMfmaCts * (JS > NPAD ? JS : NPAD) + 1;
    const int ngroups = (ncts + kMacMfmaCts - 1) / kMacMfmaCts;
    // Workgroup ids go round-robin over the XCDs (id % 8).  On one XCD consecutive ids take the ciphertext groups of one
    // column block (they read the same keys) and then the NEIGHBOURING column block: the two share every 128-byte line of
    // window and result rows (8 columns = 64 bytes), so the second finds its half in that XCD's L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, cq = slot / ngroups;
    const long col0 = (long)((cq >> 1) * 16 + xcd * 2 + (cq & 1)) * kMacMfmaCols;
    const int b0 = (slot % ngroups) * kMacMfmaCts;
    {   // window tile: a thread owns column cc and the windows jl, jl + 32, ...; 8 ciphertexts x 4 windows of loads in flight
        const int cc = threadIdx.x & 7, jl = threadIdx.x >> 3;
        const u64 *src = c + (long)b0 * c_ct_stride + col0 + cc;
#pragma unroll
        for (int bh = 0; bh < kMacMfmaCts; bh += 8) {
            u64 v[8][4];
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int j = jl + 32 * m;
                    v[b][m] = (j < k && b0 + bh + b < ncts) ? src[(long)(bh + b) * c_ct_stride + (long)j * L] : kDigC;     // kDigC: digits 0
                }
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int j = jl + 32 * m;
                    if (j < k) sa[cc * CS + (bh + b) * JS + j] = (b0 + bh + b < ncts) ? signed_digits(v[b][m]) : 0ull;
                }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int ntask = 2 * npt;                                      // a wave: 2 of the 8 columns x the prime tiles
    const long blk = (long)8 * G.lb_bytes;                          // bytes of a (column, prime tile) block
    auto keys_of = [&](int task) { return ekd + ((col0 + 2 * wave + task / npt) * G.npt + task % npt) * blk; };
    MacFrag<NFULL, TAIL> A, B0, B1;
    mac_load_keys<NFULL, TAIL>(B0, keys_of(0), G.lb_bytes, lane, G.tail_groups);
    auto build_rows = [&](int cc) {                                  // first operand: row = ciphertext n, digits la of 16 / 8 windows
        const u64 *row = sa + cc * CS + n * JS;
#pragma unroll
        for (int s = 0; s < NFULL; ++s) {
            u64 w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int j = s * 64 + g * 16 + t; w[t] = j < k ? row[j] : 0ull; }
#ifdef CUHE_MAC_PACK_DIGIT                                              // the round-2 form (A/B: profiles/r06_mac_perm_ab.txt)
#pragma unroll
            for (int la = 0; la < 8; ++la)
#pragma unroll
                for (int x = 0; x < 4; ++x) A.f[la][s][x] = (int)pack_digit(w[4 * x], w[4 * x + 1], w[4 * x + 2], w[4 * x + 3], la);
#else
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                u32 o[8];
                transpose_digits(o, w[4 * x], w[4 * x + 1], w[4 * x + 2], w[4 * x + 3]);
#pragma unroll
                for (int la = 0; la < 8; ++la) A.f[la][s][x] = (int)o[la];
            }
#endif
        }
        if (TAIL == 32) {
            u64 w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { const int j = NFULL * 64 + g * 8 + t; w[t] = j < k ? row[j] : 0ull; }
#ifdef CUHE_MAC_PACK_DIGIT
#pragma unroll
            for (int la = 0; la < 8; ++la)
                A.t32[la] = (long)((u64)pack_digit(w[0], w[1], w[2], w[3], la) | (u64)pack_digit(w[4], w[5], w[6], w[7], la) << 32);
#else
            u32 lo[8], hi[8];
            transpose_digits(lo, w[0], w[1], w[2], w[3]);
            transpose_digits(hi, w[4], w[5], w[6], w[7]);
#pragma unroll
            for (int la = 0; la < 8; ++la) A.t32[la] = (long)((u64)lo[la] | (u64)hi[la] << 32);
#endif
        }
        if (TAIL == 64) {
            u64 w[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { const int j = NFULL * 64 + g * 16 + t; w[t] = j < k ? row[j] : 0ull; }
#ifdef CUHE_MAC_PACK_DIGIT
#pragma unroll
            for (int la = 0; la < 8; ++la)
#pragma unroll
                for (int x = 0; x < 4; ++x) A.t64[la][x] = (int)pack_digit(w[4 * x], w[4 * x + 1], w[4 * x + 2], w[4 * x + 3], la);
#else
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                u32 o[8];
                transpose_digits(o, w[4 * x], w[4 * x + 1], w[4 * x + 2], w[4 * x + 3]);
#pragma unroll
                for (int la = 0; la < 8; ++la) A.t64[la][x] = (int)o[la];
            }
#endif
        }
    };
    auto run = [&](int task, const MacFrag<NFULL, TAIL> &B) {
        v4i acc[15];
#pragma unroll
        for (int t = 0; t < 15; ++t) acc[t] = v4i{kDigBias, kDigBias, kDigBias, kDigBias};
#pragma unroll
        for (int lb = 0; lb < 8; ++lb) {
#pragma unroll
            for (int s = 0; s < NFULL; ++s)
#pragma unroll
                for (int la = 0; la < 8; ++la)
                    acc[la + lb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.f[la][s], B.f[lb][s], acc[la + lb], 0, 0, 0);
            if (TAIL == 32) {
#pragma unroll
                for (int la = 0; la < 8; ++la)
                    acc[la + lb] = __builtin_amdgcn_mfma_i32_16x16x32_i8(A.t32[la], B.t32[lb], acc[la + lb], 0, 0, 0);
            }
            if (TAIL == 64) {
#pragma unroll
                for (int la = 0; la < 8; ++la)
                    acc[la + lb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A.t64[la], B.t64[lb], acc[la + lb], 0, 0, 0);
            }
        }
        // result tile: column = prime slot n, rows 4 g + r = ciphertexts.  It goes to the LDS slice of this column (the
        // window words there are in registers by now; only this wave uses the slice) as [ciphertext][prime], and leaves
        // the workgroup as 64-byte row segments after the last task
        u64 *out = sa + (2 * wave + task / npt) * CS + (task % npt) * 16 + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u32 E[15];
#pragma unroll
            for (int t = 0; t < 15; ++t) E[t] = (u32)acc[t][r];
            out[(4 * g + r) * NPAD] = fold_digits(E);
        }
    };
    for (int task = 0; task < ntask; task += 2) {                    // keys of the next task are in flight during the products of this one
        if (task + 1 < ntask) mac_load_keys<NFULL, TAIL>(B1, keys_of(task + 1), G.lb_bytes, lane, G.tail_groups);
        if (task % npt == 0) build_rows(2 * wave + task / npt);
        run(task, B0);
        if (task + 1 < ntask) {
            if (task + 2 < ntask) mac_load_keys<NFULL, TAIL>(B0, keys_of(task + 2), G.lb_bytes, lane, G.tail_groups);
            if ((task + 1) % npt == 0) build_rows(2 * wave + (task + 1) / npt);
            run(task + 1, B1);
        }
    }
    __syncthreads();
    {   // result rows leave as 64-byte segments (8 columns); a thread: column cc, primes il, il + 32, ... of every ciphertext
        const int cc = threadIdx.x & 7, il = threadIdx.x >> 3;
        u64 *o = dst + (long)b0 * dst_ct_stride + col0 + cc;
        const int nb = min(kMacMfmaCts, ncts - b0);
        for (int b = 0; b < nb; ++b)
            for (int i = il; i < np; i += kMacMfmaThreads / 8)
                __builtin_nontemporal_store(sa[cc * CS + b * NPAD + i], &o[(long)b * dst_ct_stride + (long)i * L]);
    }
}

// ---------------------------------------------------------------- relinearisation windows
// win[j][idx] = bits [w*j, w*j + w) of coefficient idx (cuhe/Base.cu:361-371), for all j < k: the raw slab is read
// ONCE, coalesced, through LDS and every window row is written coalesced, so that the k window transforms run on a
// compact u32 array (the reference re-reads the W-word coefficients with stride W for every window).
static constexpr int kWinCoef = 64, kWinGroups = 4;
static __global__ __launch_bounds__(kWinCoef * kWinGroups)
void k_extract_windows(u32 *__restrict__ win, const u32 *__restrict__ raw, int W, int w, int k, int ncoef, int clen,
                       long raw_ct_stride, long win_ct_stride) {
    raw += (long)blockIdx.y * raw_ct_stride;         // blockIdx.y: ciphertext of a batched call
    win += (long)blockIdx.y * win_ct_stride;
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [W][64]
    constexpr int CB = kWinCoef, NG = kWinGroups;
    const int ci = threadIdx.x % CB, g = threadIdx.x / CB;
    const long base = (long)blockIdx.x * CB;
    const int nvalid = (int)min((long)CB, (long)ncoef - base);
    const long slab = (long)nvalid * W;
    for (long e = threadIdx.x; e < slab; e += CB * NG) {
        const int c2 = (int)(e / W), kk = (int)(e % W);
        sh[kk * CB + c2] = raw[base * W + e];
    }
    __syncthreads();
    if (ci >= nvalid) return;
    const u32 mask = (u32)((1u << w) - 1u);
    for (int j = g; j < k; j += NG) {
        const int bit = w * j, wi = bit >> 5;
        u64 s = sh[wi * CB + ci];
        if (wi + 1 < W) s |= (u64)sh[(wi + 1) * CB + ci] << 32;
        win[(long)j * clen + base + ci] = (u32)(s >> (bit & 31)) & mask;
    }
}

// ---------------------------------------------------------------- CRT: raw -> residues (crt, Base.cu:857-879)
// residue = (sum_k word_k * (2^(32k) mod p)) mod p with a 96-bit accumulator: W multiply-adds per (coefficient, prime)
// instead of W 64-bit `%` (the reference's Horner loop, Base.cu:866-875).
// Work decomposition: a 256-thread block owns 64 coefficients; their W words are staged through LDS (coalesced
// 64*W-word slab load, transposed to [W][64]).  Wave g takes the primes i = 4g + 16t + j, j < 4, four at a time: the
// prime index is WAVE-UNIFORM, so the powers 2^(32k) mod p_i come in through scalar loads and every multiply-add is
// one v_mad_u64_u32 with an SGPR operand plus one carry add; a word is read from LDS once per four primes.
static constexpr int kCrtCoef = 64, kCrtGroups = 4, kCrtPB = 4, kCrtRow = kCrtCoef + 1;
// ACC64: the sums fit 64 bits (W 2^32 pmax <= 2^64): no carry word, one reduction
template <bool ACC64>
__global__ __launch_bounds__(kCrtCoef * kCrtGroups)
void k_crt(u32 *__restrict__ dst, const u32 *__restrict__ src, PrimeTab pt, int np, int W, int mlen, int clen,
           long src_ct_stride, long dst_ct_stride) {
    src += (long)blockIdx.y * src_ct_stride;         // blockIdx.y: polynomial of a batched call (strides in words)
    dst += (long)blockIdx.y * dst_ct_stride;
    extern __shared__ __attribute__((aligned(16))) u32 sh[];   // [W8][65], W8 = W rounded up to 8, tail rows zero; rows padded: the transposing stores spread over the banks
    constexpr int CB = kCrtCoef, NG = kCrtGroups, PB = kCrtPB, RS = kCrtRow;
    const int ci = threadIdx.x % CB;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x / CB);
    const int W8 = (W + 7) & ~7;
    const long base = (long)blockIdx.x * CB;
    const int nvalid = (int)min((long)CB, (long)mlen - base);
    const long slab = (long)nvalid * W;
    for (int e = threadIdx.x; e < (W8 - W) * RS; e += CB * NG) sh[W * RS + e] = 0;
    {   // (coefficient, word) of element e advance without a division per element
        const int dc = (CB * NG) / W, dk = (CB * NG) % W;
        int c2 = (int)threadIdx.x / W, k = (int)threadIdx.x % W;
        for (long e = threadIdx.x; e < slab; e += CB * NG) {
            sh[k * RS + c2] = src[base * W + e];
            c2 += dc; k += dk;
            if (k >= W) { k -= W; ++c2; }
        }
    }
    __syncthreads();
    if (ci >= nvalid) return;
    // the table has maxW (a multiple of 8) words per row and kCrtPB zero rows after the last prime: no guards below
    for (int i0 = g * PB; i0 < np; i0 += NG * PB) {
        u64 lo[PB]; u32 hi[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) { lo[j] = 0; hi[j] = 0; }
        const u32 *pw = pt.pow32 + (long)i0 * pt.maxW;
        for (int k0 = 0; k0 < W8; k0 += 8) {
            u32 c[PB][8];
#pragma unroll
            for (int j = 0; j < PB; ++j)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) c[j][kk] = pw[(long)j * pt.maxW + k0 + kk];   // uniform: s_load_dwordx8
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const u32 x = sh[(k0 + kk) * RS + ci];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const u64 nl = (u64)x * c[j][kk] + lo[j];
                    if (!ACC64) hi[j] += (nl < lo[j]);
                    lo[j] = nl;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int i = i0 + j;
            if (i < np) {
                const u32 p = pt.p[i];
                const u64 m = pt.pinv[i];
                const u32 r1 = mod_small(lo[j], p, m);
                if (ACC64) dst[(long)i * clen + base + ci] = r1;
                else {
                    const u64 r2 = (u64)hi[j] * pt.e64[i] + r1;
                    dst[(long)i * clen + base + ci] = mod_small(r2, p, m);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- ICRT: residues -> raw (icrt, Base.cu:880-924)
// value = sum_i ((x_i*b_i) mod p_i) * m_i  mod M, in [0, M).  Instead of a conditional multiword
// subtract after every prime (104-word register array) the quotient q = floor(sum_i t_i/p_i) is
// estimated in f64, q*M is folded into the column sums, and a single +-M fix-up pass makes the
// result exact whatever the rounding of the estimate.
struct IcrtTab {
    const u32 *M;       // [W]
    const u32 *mi;      // [np8][W4]  m_i = M / p_i, rows padded with zeros to W4 = ceil4(W), zero rows up to np8 = ceil8(np)
    const u32 *bi;      // [np]
    const double *rp;   // [np] 1/p_i
};
// Work decomposition: a 256-thread block owns 64 coefficients, wave g = tid/64.
//   phase 1: wave g forms the residue products t_i = ((x_i mod p_i) b_i) mod p_i of the primes i = g mod 4 (the prime
//            is wave-uniform: p_i, b_i, 1/p_i are scalars) into LDS and its share of alpha = sum t_i / p_i;
//   phase 2: wave g owns the output words k = 4g + 16t + j, j < 4: the 96-bit column sums  sum_i t_i * m_i[k]  take
//            one LDS read of t_i per FOUR multiply-adds, the m_i words arrive through scalar loads (8 primes x 4
//            words per block, unguarded thanks to the zero padding) and a multiply-add is one v_mad_u64_u32 with an
//            SGPR operand plus a carry add; q*M is folded in, q = floor(alpha).  The wave then adds its four columns up
//            (they sit 32 bits apart) into four result words and ONE signed 64-bit carry towards the next block of words,
//            so that only 24 bytes per coefficient and block go through LDS;
//   phase 3: wave 0 ripples the block carries (W/4 steps instead of W) and applies the rare +-M fix-up; the block stores
//            its 64*W-word slab coalesced.
// 28 KB of LDS at 48 primes (5 workgroups per CU); the first version kept every column in LDS (70 KB, 2 per CU) and
// rippled W words on one wave.  (The reference runs one thread per coefficient with a 104-word register array, Base.cu:884.)
static constexpr int kIcrtCoef = 64, kIcrtGroups = 4, kIcrtKB = 4;
static inline size_t icrt_lds_bytes(int np, int W) {
    const size_t np8 = (size_t)(np + 7) & ~(size_t)7, nb = (size_t)(W + kIcrtKB - 1) / kIcrtKB;
    return (size_t)kIcrtCoef * (np8 * 4 + nb * 24 + kIcrtGroups * 8);
}
// four 96-bit multiply-adds  (hi:lo)[j] += t * c_j  with wave-uniform c_j (SGPRs): the multiply-add's own carry-out feeds
// the add of the top word -- 2 instructions per multiply-add (the compiler's form: multiply, 64-bit add, compare, select,
// add = 5).  The four chains are interleaved so that three instructions lie between a write of a carry (SGPR pair / VCC)
// and the VALU read of it (VALUWriteSGPRVALURead, tools/asm_hazard_check.py).
__device__ __forceinline__ void icrt_mac4(u32 t, u32 c0, u32 c1, u32 c2, u32 c3, u64 (&lo)[4], u32 (&hi)[4]) {
    u64 sA, sB, sC;
    asm("v_mad_u64_u32 %[l0], vcc, %[t], %[c0], %[l0]\n\t"
        "v_mad_u64_u32 %[l1], %[sA], %[t], %[c1], %[l1]\n\t"
        "v_mad_u64_u32 %[l2], %[sB], %[t], %[c2], %[l2]\n\t"
        "v_mad_u64_u32 %[l3], %[sC], %[t], %[c3], %[l3]\n\t"
        "v_addc_co_u32_e32 %[h0], vcc, 0, %[h0], vcc\n\t"
        "v_addc_co_u32_e64 %[h1], %[sA], 0, %[h1], %[sA]\n\t"
        "v_addc_co_u32_e64 %[h2], %[sB], 0, %[h2], %[sB]\n\t"
        "v_addc_co_u32_e64 %[h3], %[sC], 0, %[h3], %[sC]"
        : [l0] "+v"(lo[0]), [l1] "+v"(lo[1]), [l2] "+v"(lo[2]), [l3] "+v"(lo[3]),
          [h0] "+v"(hi[0]), [h1] "+v"(hi[1]), [h2] "+v"(hi[2]), [h3] "+v"(hi[3]),
          [sA] "=&s"(sA), [sB] "=&s"(sB), [sC] "=&s"(sC)
        : [t] "v"(t), [c0] "s"(c0), [c1] "s"(c1), [c2] "s"(c2), [c3] "s"(c3)
        : "vcc");
}
// the same with 64-bit accumulators, for levels where a whole column sum fits them -- np (2^logCrtPrime + 1) < 2^32: every
// t_i < p_i < 2^logCrtPrime, every word below 2^32, q < np -- which is every parameter set of the reference's examples (24- and
// 25-bit primes): ONE instruction per multiply-add, the carry-out is dead
__device__ __forceinline__ void icrt_mac4_64(u32 t, u32 c0, u32 c1, u32 c2, u32 c3, u64 (&lo)[4]) {
    asm("v_mad_u64_u32 %[l0], vcc, %[t], %[c0], %[l0]\n\t"
        "v_mad_u64_u32 %[l1], vcc, %[t], %[c1], %[l1]\n\t"
        "v_mad_u64_u32 %[l2], vcc, %[t], %[c2], %[l2]\n\t"
        "v_mad_u64_u32 %[l3], vcc, %[t], %[c3], %[l3]"
        : [l0] "+v"(lo[0]), [l1] "+v"(lo[1]), [l2] "+v"(lo[2]), [l3] "+v"(lo[3])
        : [t] "v"(t), [c0] "s"(c0), [c1] "s"(c1), [c2] "s"(c2), [c3] "s"(c3)
        : "vcc");
}
// optional second output of k_icrt: the relinearisation windows of the coefficients (what k_extract_windows makes of the
// raw words), written straight from the result words in LDS -- the batched chain then needs neither the raw rows nor
// the extraction kernel (151 + 151 MB of traffic per 32 ciphertexts at config 4)
struct IcrtWindows { u32 *win; long ct_stride; int w, k, clen; };
// phase 3 of the ICRT kernels: wave 0 ripples the block carries and applies the +-M fix-up, then the block stores its slab
__device__ __forceinline__ void icrt_finish(u32 *__restrict__ dst, uint4 *blk, const long long *bcar, int nb, int W, const IcrtTab &it,
                                            int ci, int g, long base, int nvalid, const IcrtWindows &wo) {
    constexpr int CB = kIcrtCoef, NG = kIcrtGroups;
    u32 *out = reinterpret_cast<u32 *>(blk);         // word k of coefficient c: out[((k / 4) * CB + c) * 4 + k % 4]
    if (g == 0) {
        long long cin = 0;
        for (int b = 0; b < nb; ++b) {
            uint4 v = blk[b * CB + ci];
            long long t = (long long)v.x + cin; v.x = (u32)t; t >>= 32;
            t += (long long)v.y; v.y = (u32)t; t >>= 32;
            t += (long long)v.z; v.z = (u32)t; t >>= 32;
            t += (long long)v.w; v.w = (u32)t; t >>= 32;
            blk[b * CB + ci] = v;
            cin = t + bcar[b * CB + ci];
        }
        // S - q*M lies in (-M, 2M) and M < 2^(32W): everything above word W-1 (the zero-padded words of the last block and
        // the final carry) is its sign extension: cin < 0  <=>  negative
        auto word = [&](int k) -> u32 & { return out[((k >> 2) * CB + ci) * 4 + (k & 3)]; };
        int fix = 0;                          // +1: add M, -1: subtract M
        if (cin < 0) fix = 1;
        else {
            bool ge = true;                   // out >= M ?
            for (int k = W - 1; k >= 0; --k) {
                const u32 x = word(k), y = it.M[k];
                if (x != y) { ge = x > y; break; }
            }
            if (ge) fix = -1;
        }
        if (fix != 0) {
            long long cy = 0;
            for (int k = 0; k < W; ++k) {
                const long long t = (long long)word(k) + (long long)fix * (long long)it.M[k] + cy;
                word(k) = (u32)t;
                cy = t >> 32;
            }
        }
    }
    __syncthreads();
    if (wo.win && ci < nvalid) {                      // win[j][coefficient] = bits [w j, w j + w) (cuhe/Base.cu:361-371); a wave: windows g, g + 4, ...
        u32 *wrow = wo.win + (long)blockIdx.y * wo.ct_stride + base + ci;
        const u32 mask = (u32)((1u << wo.w) - 1u);
        for (int j = g; j < wo.k; j += NG) {
            const int bit = wo.w * j, wi = bit >> 5;
            u64 sv = out[((wi >> 2) * CB + ci) * 4 + (wi & 3)];
            if (wi + 1 < W) sv |= (u64)out[(((wi + 1) >> 2) * CB + ci) * 4 + ((wi + 1) & 3)] << 32;
            wrow[(long)j * wo.clen] = (u32)(sv >> (bit & 31)) & mask;
        }
    }
    if (!dst) return;
    const int slab = nvalid * W, dc = (CB * NG) / W, dk = (CB * NG) % W;       // coalesced: (coefficient, word) advance without a division per element
    int c2 = (int)threadIdx.x / W, k = (int)threadIdx.x % W;
    for (int e = threadIdx.x; e < slab; e += CB * NG) {
        dst[base * W + e] = out[((k >> 2) * CB + c2) * 4 + (k & 3)];
        c2 += dc; k += dk;
        if (k >= W) { k -= W; ++c2; }
    }
}
template <bool ACC64>
__global__ __launch_bounds__(kIcrtCoef * kIcrtGroups)
void k_icrt(u32 *__restrict__ dst, const u32 *__restrict__ src, PrimeTab pt, IcrtTab it,
            int np, int W, int mlen, int clen, long src_ct_stride, long dst_ct_stride, IcrtWindows wo) {
    src += (long)blockIdx.y * src_ct_stride;         // blockIdx.y: ciphertext of a batched call (strides in words)
    if (dst) dst += (long)blockIdx.y * dst_ct_stride;
    extern __shared__ __attribute__((aligned(16))) unsigned char shraw[];
    constexpr int CB = kIcrtCoef, NG = kIcrtGroups, KB = kIcrtKB;
    const int np8 = (np + 7) & ~7, W4 = (W + 3) & ~3, nb = W4 / KB;
    uint4 *blk = reinterpret_cast<uint4 *>(shraw);                       // [nb][CB] four result words of a block (before the incoming carry)
    long long *bcar = reinterpret_cast<long long *>(blk + (size_t)nb * CB);   // [nb][CB] carry out of the block
    double *alphaP = reinterpret_cast<double *>(bcar + (size_t)nb * CB); // [NG][CB]
    u32 *tt = reinterpret_cast<u32 *>(alphaP + NG * CB);                 // [np8][CB]
    const int ci = threadIdx.x % CB;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x / CB);
    const long base = (long)blockIdx.x * CB;
    const int nvalid = (int)min((long)CB, (long)mlen - base);
    const bool live = ci < nvalid;
    double a = 0.0;
    // phase 1 is straight-line per block of PU primes (clamped index, results masked): the per-prime constants are
    // wave-uniform scalar loads, and a branch per prime made every one of them wait for the previous (scalar-latency bound)
    constexpr int PU = 8;
    for (int i0 = g; i0 < np8; i0 += NG * PU) {
        u32 xr[PU], pp[PU], bb[PU]; u64 mm[PU]; double rr[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int ic = min(i0 + NG * u, np - 1);
            pp[u] = pt.p[ic]; mm[u] = pt.pinv[ic]; bb[u] = it.bi[ic]; rr[u] = it.rp[ic];
            xr[u] = live ? src[(long)ic * clen + base + ci] : 0u;
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int i = i0 + NG * u;
            u32 v = mod_small((u64)xr[u] * bb[u], pp[u], mm[u]);          // (x mod p) b mod p = x b mod p: x < 2^32, b < p < 2^31
            if (i >= np) v = 0;
            a += (double)v * rr[u];
            if (i < np8) tt[i * CB + ci] = v;
        }
    }
    alphaP[g * CB + ci] = a;
    __syncthreads();
    double alpha = 0.0;
#pragma unroll
    for (int gg = 0; gg < NG; ++gg) alpha += alphaP[gg * CB + ci];
    const u32 q = (u32)alpha;            // floor; may be off by one either way -> fixed below
    typedef __int128 i128;
    for (int k0 = g * KB; k0 < W4; k0 += NG * KB) {
        u64 lo[KB]; u32 hi[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) { lo[j] = 0; hi[j] = 0; }
        for (int i0 = 0; i0 < np8; i0 += 8) {
            u32 c[8][KB];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii)
#pragma unroll
                for (int j = 0; j < KB; ++j) c[ii][j] = it.mi[(long)(i0 + ii) * W4 + k0 + j];      // uniform: s_load_dwordx4
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const u32 t = tt[(i0 + ii) * CB + ci];
                if constexpr (ACC64) icrt_mac4_64(t, c[ii][0], c[ii][1], c[ii][2], c[ii][3], lo);
                else icrt_mac4(t, c[ii][0], c[ii][1], c[ii][2], c[ii][3], lo, hi);
            }
        }
        // the four columns, 32 bits apart, minus q * M: four words and a signed carry (column < 2^71, so the carry fits)
        u32 w[KB];
        long long carry = 0;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int k = k0 + j;
            const u64 qm = k < W ? (u64)q * it.M[k] : 0;
            const i128 col = (i128)(unsigned __int128)lo[j] + ((i128)hi[j] << 64) - (i128)(unsigned __int128)qm + (i128)carry;
            w[j] = (u32)col;
            carry = (long long)(col >> 32);
        }
        blk[(k0 / KB) * CB + ci] = make_uint4(w[0], w[1], w[2], w[3]);
        bcar[(k0 / KB) * CB + ci] = carry;
    }
    __syncthreads();
    icrt_finish(dst, blk, bcar, nb, W, it, ci, g, base, nvalid, wo);
}

// ---- gather / scatter of equally sized blocks through a pointer list passed BY VALUE (no table in device memory): the
// gate scheduler of the C++ layer (cuhe_amd/cxx/Scheduler.h) runs ready gates of one kind as ONE batched call on contiguous
// arrays, while every ciphertext of the client owns its own block.  16 bytes per lane and step; bytes is a multiple of 16.
constexpr int kPtrListMax = 64;       // 512 bytes per list; kernels take up to three (the kernel-argument segment holds 4 KB)
struct PtrList { void *p[kPtrListMax]; };
// elementwise gates on LISTS of separately owned ciphertexts (blockIdx.y / z = item): z[i] = x[i] (*|+) y[i] on ct rows, z[i] = (a[i] + b[i]) mod p on CRT rows
template <bool MUL>
__global__ __launch_bounds__(256)
void k_ntt_binop_list(PtrList zl, PtrList xl, PtrList yl, long n2) {
    ulonglong2 *z = (ulonglong2 *)zl.p[blockIdx.y];
    const ulonglong2 *x = (const ulonglong2 *)xl.p[blockIdx.y], *y = (const ulonglong2 *)yl.p[blockIdx.y];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        const ulonglong2 a = x[i], b = y[i];
        ulonglong2 r;
        if (MUL) { r.x = mulp(a.x, b.x); r.y = mulp(a.y, b.y); }
        else     { r.x = addp(a.x, b.x); r.y = addp(a.y, b.y); }
        z[i] = r;
    }
}
template <int VEC>          // coefficients per thread: 4 (16-byte accesses; mlen, clen multiples of 4, blocks 16-byte aligned) or 1
__global__ __launch_bounds__(256)
void k_crt_add_list(PtrList zl, PtrList al, PtrList bl, PrimeTab pt, int mlen, int clen) {
    const int crt = blockIdx.y, idx = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (idx >= mlen) return;
    const long o = (long)crt * clen + idx;
    u32 *z = (u32 *)zl.p[blockIdx.z];
    const u32 *a = (const u32 *)al.p[blockIdx.z], *b = (const u32 *)bl.p[blockIdx.z];
    const u32 p = pt.p[crt];
    const u64 m = pt.pinv[crt];
    if (VEC == 4) {
        const uint4 x = *reinterpret_cast<const uint4 *>(a + o), y = *reinterpret_cast<const uint4 *>(b + o);
        *reinterpret_cast<uint4 *>(z + o) = make_uint4(mod_small((u64)x.x + y.x, p, m), mod_small((u64)x.y + y.y, p, m),
                                                       mod_small((u64)x.z + y.z, p, m), mod_small((u64)x.w + y.w, p, m));
    } else z[o] = mod_small((u64)a[o] + b[o], p, m);
}
// modSwitch over a list of separately owned ciphertexts (blockIdx.z = item): dst[i] may be src[i] (row r only depends on rows r and
// np - 1, and row np - 1 is never written), which is how the gate scheduler switches CRT-domain ciphertexts in their own blocks
template <int VEC>
__global__ __launch_bounds__(256)
void k_modswitch_list(PtrList dl, PtrList sl, PrimeTab pt, const u32 *__restrict__ invp, int np, int mlen, int clen, int modmsg) {
    modswitch_body<VEC>((u32 *)dl.p[blockIdx.z], (const u32 *)sl.p[blockIdx.z], pt, invp, np, mlen, clen, modmsg);
}
// cNot over a list (blockIdx.y = item): z[i] = x[i] with (modMsg - 1) added to the constant coefficient of every row (crt_add_int,
// Base.cu:1096-1100); z[i] == x[i]: only the np constant terms are touched, otherwise the other coefficients are copied as well
static __global__ __launch_bounds__(256)
void k_crt_add_int_list(PtrList zl, PtrList xl, unsigned a, PrimeTab pt, int np, int mlen, int clen) {
    u32 *z = (u32 *)zl.p[blockIdx.y];
    const u32 *x = (const u32 *)xl.p[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (z != x) {
        const long total = (long)np * mlen;
        for (long e = t; e < total; e += (long)gridDim.x * blockDim.x) {
            const int row = (int)(e / mlen), idx = (int)(e - (long)row * mlen);
            const long o = (long)row * clen + idx;
            u32 v = x[o];
            if (idx == 0) { const u32 p = pt.p[row]; const u64 m = pt.pinv[row]; v = mod_small((u64)v + mod_small(a, p, m), p, m); }
            z[o] = v;
        }
    } else if (t < np) {
        const u32 p = pt.p[t]; const u64 m = pt.pinv[t];
        z[(long)t * clen] = mod_small((u64)x[(long)t * clen] + mod_small(a, p, m), p, m);
    }
}
static __global__ __launch_bounds__(256)
void k_copy_list(PtrList dl, PtrList sl, long bytes) {
    char *d = (char *)dl.p[blockIdx.y];
    const char *sr = (const char *)sl.p[blockIdx.y];
    for (long o = ((long)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (long)gridDim.x * 256 * 16)
        *(v4i *)(d + o) = __builtin_nontemporal_load((const v4i *)(sr + o));
}
template <bool GATHER>
__global__ __launch_bounds__(256)
void k_move_blocks(char *__restrict__ contig, PtrList list, long bytes) {
    char *blk = (char *)list.p[blockIdx.y];
    char *row = contig + (long)blockIdx.y * bytes;
    for (long o = ((long)blockIdx.x * 256 + threadIdx.x) * 16; o < bytes; o += (long)gridDim.x * 256 * 16) {
        if (GATHER) *(v4i *)(row + o) = __builtin_nontemporal_load((const v4i *)(blk + o));
        else *(v4i *)(blk + o) = __builtin_nontemporal_load((const v4i *)(row + o));
    }
}

}  // namespace cuhe
