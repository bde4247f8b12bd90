// modp.cuh -- device arithmetic in the field Z_P, P = 2^64 - 2^32 + 1, for gfx950.
//
// Replaces cuhe/ModP.h:151-289 of the reference (inline-PTX carry chains) with
// wave64-friendly 32/64-bit integer code the AMDGPU backend lowers to
// v_add_co/v_addc_co/v_mad_u64_u32.  All results are CANONICAL residues in
// [0, P) (the reference's `> valP` compares can return P for 0 -- SURVEY A.7 --
// which would corrupt the later `% p_i`; we do not reproduce that).
//
// Identities used: phi = 2^32, phi^2 = phi - 1, phi^3 = -1 (mod P), so 2 has
// order 192 and 8 = 2^3 is a primitive 64-th root of unity: every butterfly
// inside a <=64-point sub-transform is an add/sub plus a shift.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cuhe {

typedef unsigned long long u64;
typedef unsigned int u32;

static constexpr u64 kP = 0xffffffff00000001ULL;
static constexpr u64 kEps = 0xffffffffULL;   // 2^64 mod P

// canonical - canonical -> canonical.  a - b, then "+ P on borrow" without a select: + P = - eps = + 1 - 2^32, so with
// the borrow B as a lane mask the low word takes B as a carry-in (lo + B, carry C) and the high word loses B & ~C
// (d >= 2^32 whenever there was a borrow, so the high word cannot underflow): 4 VALU + 1 SALU, against 6 VALU for the
// compare-and-select form.  The carries travel in SGPR pairs; LLVM pads "VALU writes an SGPR pair / VCC -> VALU reads it"
// with two wait states on gfx940/950 (GCNHazardRecognizer, VALUWriteSGPRVALURead) and nothing pads the inside of an asm
// string, so the two places where that happens carry their own s_nop 1 (tools/asm_hazard_check.py verifies every asm
// site of the generated code, VCC included).  Chosen by A/B on the 64K transform (profiles/r02_field_arith_ab.txt):
// same speed as the unpadded chain, faster than every SALU-free formulation.
__device__ __forceinline__ u64 subp(u64 a, u64 b) {
    if (__builtin_constant_p(b) && b == 0) return a;     // e.g. the bits above 2^96 of a shifted 32-bit sample: the asm below would hide the zero
    u32 lo, hi; u64 bw, t;
    asm("v_sub_co_u32_e64 %0, %2, %4, %6\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %1, %2, %5, %7, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %0, %3, %0, 0, %2\n\t"
        "s_andn2_b64 %2, %2, %3\n\t"
        "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"
        : "=&v"(lo), "=&v"(hi), "=&s"(bw), "=&s"(t)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "scc");
    u64 d = ((u64)hi << 32) | lo;
    asm("" : "+v"(d));              // keep the two words a register PAIR: otherwise 64-bit shifts of d are split into word operations
    return d;
}
// canonical + canonical -> canonical: "the 64-bit sum carried" and "the sum is >= P" exclude each other and both call
// for + eps, applied as ONE v_mad_u64_u32 with a 0/1 flag.  (A single-compare form exists -- t = a + b + eps (mod 2^64)
// is the result exactly when t < a -- but needs a two-word select, and two v_cndmask reading one condition issue at
// less than half rate on gfx950: tools/ubench_rates.hip, profiles/r02_valu_cost_model.txt.)
__device__ __forceinline__ u64 addp(u64 a, u64 b) {
    u64 s = a + b;
    const u32 f = ((s < a) | (s >= kP)) ? 1u : 0u;
    return (u64)f * 0xffffffffu + s;
}
__device__ __forceinline__ u64 negp(u64 a) { return a ? kP - a : 0; }

// any u64 -> canonical
__device__ __forceinline__ u64 canon(u64 r) {
    u64 t = r + kEps;
    return (t < r) ? t : r;
}

// lo + m*eps for a 32-bit m, canonical result: ONE correction, because "the 64-bit sum carried" and "the sum is >= P"
// exclude each other and both call for + eps.  Only the multiply-add is asm: its carry-out lands in an SGPR pair (a
// lane mask), which inverse_ballot hands back to the compiler as a per-lane boolean; the OR with the >= P compare, the
// select and the final add are ordinary code the compiler schedules (and pads) itself.  The mask is consumed by the
// s_or_b64 of that OR -- an SALU read, interlocked -- never directly by a VALU instruction.
__device__ __forceinline__ u64 mad_eps(u32 m, u64 lo) {
    if (__builtin_constant_p(m) && m == 0) return canon(lo);      // small shifts of 32-bit samples: nothing above 2^64 (canon folds too)
    u64 r, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(r), "=s"(carry) : "v"(m), "v"(0xffffffffu), "v"(lo));
    const bool f = __builtin_amdgcn_inverse_ballot_w64(carry) | (r >= kP);
    return (u64)(f ? 1u : 0u) * 0xffffffffu + r;
}

// 128-bit (hi:lo) -> canonical; hi = hh:hl.  lo + hl*(phi-1) - hh
__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 r = mad_eps(hl, lo);
    return subp(r, (u64)hh);           // r - hh, + P on borrow (hh < 2^32 is canonical)
}

// canonical x canonical -> canonical.  Chained partial products: every 64-bit addend "upper word of the previous product,
// zero extended" is assembled with a v_mov into a pair whose upper register holds zero -- 5 moves per product, 13.6 of the
// 154 instructions per point of the 64K transform.  A move-free form (x = a0 b0, y = a1 b1, the 65-bit cross term on one
// multiply-add with its carry in an SGPR pair, word sums in place: 8 instead of 10 instructions, 150.8 per point) was built in
// round 4 and is SLOWER on the transforms that matter (2.73 vs 2.78 M/s at 64K, 6.70 vs 6.88 at 32K, same box, alternating:
// profiles/r04_mulp_ab.txt): v_mov_b32 is one of the plain 32-bit instructions that issue at 1.5-1.7x the rate of the
// carry / 64-bit ones that replaced it (profiles/r02_valu_cost_model.txt).  Instruction COUNT is not the cost model here.
__device__ __forceinline__ u64 mulp(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 t = (u64)a0 * b0;
    u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v = (u64)a1 * b0 + (u32)u;
    u64 w = (u64)a1 * b1 + (u >> 32) + (v >> 32);
    u64 lo = (v << 32) | (u32)t;
    return reduce128(lo, w);
}

// canonical u64 times a u32 -> canonical  (96-bit product)
__device__ __forceinline__ u64 mulp_u32(u64 a, u32 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32);
    u64 t = (u64)a0 * b;
    u64 u = (u64)a1 * b + (t >> 32);
    u64 lo = (u << 32) | (u32)t;
    return mad_eps((u32)(u >> 32), lo);            // bits 64..95 fold with eps
}

// x * 2^K mod P for a compile-time K in [0, 96); x canonical.
//   x*2^K = lo + mid*2^64 + hi*2^96  ==  lo + mid*eps - hi
template <int K>
__device__ __forceinline__ u64 shlp(u64 x) {
    static_assert(K >= 0 && K < 96, "shift out of range");
    if constexpr (K == 0) {
        return x;
    } else if constexpr (K < 32) {
        return mad_eps((u32)(x >> (64 - K)), x << K);
    } else if constexpr (K == 32) {
        return mad_eps((u32)(x >> 32), (u64)(u32)x << 32);         // x0*phi + x1*(phi-1)
    } else if constexpr (K < 64) {
        u64 r = mad_eps((u32)(x >> (64 - K)), x << K);      // bits 0..95
        return subp(r, x >> (96 - K));                       // minus bits 96.. (2^96 = -1)
    } else if constexpr (K == 64) {
        u32 x0 = (u32)x, x1 = (u32)(x >> 32);
        u64 t1 = ((u64)x0 << 32) - x0;             // x0 * eps
        return subp(t1, (u64)x1);
    } else {
        u32 mid = (u32)(x << (K - 64));            // bits 64..95
        u64 hi = x >> (96 - K);
        u64 t1 = ((u64)mid << 32) - mid;
        return subp(t1, hi);
    }
}

// (u - v) * 2^K for K in [0,192): K >= 96 folds the sign into the subtraction.
template <int K>
__device__ __forceinline__ u64 sub_shlp(u64 u, u64 v) {
    if constexpr (K >= 96) return shlp<K - 96>(subp(v, u));
    else return shlp<K>(subp(u, v));
}

// canonical value mod a 32-bit prime p; m = floor(2^64 / p)
__device__ __forceinline__ u32 mod_small(u64 x, u32 p, u64 m) {
    u64 q = __umul64hi(x, m);
    u64 r = x - q * p;
    if (r >= p) r -= p;
    return (u32)r;
}

// ---------------------------------------------------------------------------
// In-register N-point DFT over Z_P with root 2^(192/N) (N | 64), natural order
// in and out.  Radix-2 DIF with compile-time shifts; the final bit reversal is
// register renaming.  Replaces _ntt4/_ntt8/_ntt8_ext + the 8x8 LDS transposes
// of cuhe/Base.cu:225-306.
// ---------------------------------------------------------------------------
template <int N, int LEN, int BASE, int J>
__device__ __forceinline__ void dif_pair(u64 (&x)[N]) {
    constexpr int H = LEN / 2;
    constexpr int K = (192 / LEN) * J;
    u64 u = x[BASE + J], v = x[BASE + J + H];
    x[BASE + J] = addp(u, v);
    x[BASE + J + H] = sub_shlp<K>(u, v);
}
template <int N, int LEN, int BASE, int J>
struct DifBlock {
    static __device__ __forceinline__ void run(u64 (&x)[N]) {
        dif_pair<N, LEN, BASE, J>(x);
        if constexpr (J + 1 < LEN / 2) DifBlock<N, LEN, BASE, J + 1>::run(x);
    }
};
template <int N, int LEN, int BASE>
struct DifStage {
    static __device__ __forceinline__ void run(u64 (&x)[N]) {
        DifBlock<N, LEN, BASE, 0>::run(x);
        if constexpr (BASE + LEN < N) DifStage<N, LEN, BASE + LEN>::run(x);
    }
};
template <int N, int LEN>
struct DifAll {
    static __device__ __forceinline__ void run(u64 (&x)[N]) {
        DifStage<N, LEN, 0>::run(x);
        if constexpr (LEN > 2) DifAll<N, LEN / 2>::run(x);
    }
};
template <int N>
__host__ __device__ constexpr int bitrev(int i) {
    int r = 0;
    for (int b = 1; b < N; b <<= 1) { r = (r << 1) | (i & 1); i >>= 1; }
    return r;
}
// y[k] = sum_j x[j] * (2^(192/N))^(j*k); result left in x with x[bitrev(k)] = y[k].
template <int N>
__device__ __forceinline__ void dft_bitrev(u64 (&x)[N]) { DifAll<N, N>::run(x); }

}  // namespace cuhe
