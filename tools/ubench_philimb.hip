// tools/ubench_philimb.hip -- experiment (a) of VERDICT r02: the in-register 16-point transform on REDUNDANT phi-radix limbs
// (a = a0 + a1 phi, phi = 2^32, phi^2 = phi - 1, a0 / a1 signed 64-bit, no carries and no canonical fix-ups inside the
// butterflies) against the canonical form of modp.cuh (dft_regs<16>), both on register-resident data, same launch
// geometry; results compared on the device's output.
//   add / sub      : limb-wise 64-bit add / subtract
//   * 2^(32 q)     : (a0, a1) -> (-a1, a0 + a1) [q = 1], (-(a0 + a1), a0) [q = 2]; 2^96 = -1
//   * 2^s, s < 32  : limb-wise shift; the limbs GROW by s bits
//   normalise      : a0 = l0 + h0 phi, a1 = l1 + h1 phi  ->  (l0 - h1) + (l1 + h0 + h1) phi   (limbs back to ~33 bits)
// Bit growth: a butterfly output that is shifted gains up to 30 bits (root 2^12: s in {0, 12, 24, 4, 16, 28, 8, 20}) plus 2
// for the subtraction and the phi rotation, so from 34-bit limbs it can take ONE shift before it has to be normalised: the
// normalisation cannot wait for the LDS exchange, it follows every shifted output.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_philimb.hip -o tools/ubench_philimb
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../cuhe_amd/csrc/ntt_kernels.cuh"
using namespace cuhe;
typedef long long i64;
#define ITERS 64

struct PL { i64 a0, a1; };
__device__ __forceinline__ PL pl_from(u64 x) { return PL{(i64)(u32)x, (i64)(x >> 32)}; }
__device__ __forceinline__ PL pl_add(PL u, PL v) { return PL{u.a0 + v.a0, u.a1 + v.a1}; }
__device__ __forceinline__ PL pl_sub(PL u, PL v) { return PL{u.a0 - v.a0, u.a1 - v.a1}; }
__device__ __forceinline__ PL pl_norm(PL a) {
    const i64 l0 = (i64)(u32)a.a0, h0 = a.a0 >> 32, l1 = (i64)(u32)a.a1, h1 = a.a1 >> 32;
    return PL{l0 - h1, l1 + h0 + h1};
}
// a * 2^K, K in [0, 192); the input limbs are below 2^36 in magnitude, the result is normalised when it was shifted
template <int K>
__device__ __forceinline__ PL pl_shl(PL a) {
    constexpr int KK = K % 96, q = KK / 32, s = KK % 32;
    if constexpr (K >= 96) { a.a0 = -a.a0; a.a1 = -a.a1; }
    if constexpr (q == 1) a = PL{-a.a1, a.a0 + a.a1};
    else if constexpr (q == 2) a = PL{-(a.a0 + a.a1), a.a0};
    if constexpr (s > 0) {
        if constexpr (s > 25) a = pl_norm(a);             // 37 + s would pass 62 bits
        a = PL{a.a0 << s, a.a1 << s};
        a = pl_norm(a);
    }
    return a;
}
// canonical residue of a0 + a1 phi, limbs below 2^40 in magnitude
__device__ __forceinline__ u64 pl_canon(PL a) {
    a = pl_norm(a);                                       // |a0| < 2^33, |a1| < 2^34
    // a1 = l1 + h1 phi once more (h1 in [-4, 4)): value = (a0 - h1) + (l1 + h1) phi, then + 8 P to make both parts positive
    const i64 h1 = a.a1 >> 32, l1 = (i64)(u32)a.a1;
    const i64 lo = a.a0 - h1 + ((i64)1 << 34), hi = l1 + h1 + 16;          // lo in (0, 2^35), hi in (0, 2^33): lo + hi phi - (2^34 + 16 phi)
    // lo + hi * 2^32 = 96-bit value: fold with 2^64 = 2^32 - 1
    const u64 w = (u64)lo + ((u64)(u32)hi << 32);          // may carry
    const u32 carry = w < (u64)lo ? 1u : 0u;
    const u32 top = (u32)((u64)hi >> 32) + carry;          // bits 64.. (a few)
    u64 r = mad_eps(top, w);
    // remove the bias 2^34 + 16 phi = 2^34 + 2^36 (mod P unchanged: below P)
    return subp(r, ((u64)1 << 34) + ((u64)16 << 32));
}

template <int LEN, int BASE, int J>
__device__ __forceinline__ void pl_pair(PL (&x)[16]) {
    constexpr int H = LEN / 2, K = (192 / LEN) * J;
    const PL u = x[BASE + J], v = x[BASE + J + H];
    x[BASE + J] = pl_add(u, v);
    x[BASE + J + H] = pl_shl<K>(pl_sub(u, v));
}
template <int LEN, int BASE, int J>
struct PlBlock { static __device__ __forceinline__ void run(PL (&x)[16]) { pl_pair<LEN, BASE, J>(x); if constexpr (J + 1 < LEN / 2) PlBlock<LEN, BASE, J + 1>::run(x); } };
template <int LEN, int BASE>
struct PlStage { static __device__ __forceinline__ void run(PL (&x)[16]) { PlBlock<LEN, BASE, 0>::run(x); if constexpr (BASE + LEN < 16) PlStage<LEN, BASE + LEN>::run(x); } };
__device__ __forceinline__ void pl_dft16(PL (&x)[16]) {
    PlStage<16, 0>::run(x);
    PlStage<8, 0>::run(x);
    // sums of the first two levels are un-normalised (+2 bits), every shifted value is normalised: all limbs below 2^37
    PlStage<4, 0>::run(x);
    PlStage<2, 0>::run(x);
}

// MODE 0: canonical form; 1: phi limbs, conversion to / from canonical around EVERY transform (what an LDS exchange or a
// general multiplication needs); 2: phi limbs, conversions outside the timed loop (lower bound: butterflies only)
template <int MODE>
__global__ __launch_bounds__(256) void kern(u64 *out, u64 seed, int iters) {
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = canon(seed * (2 * i + 3) + threadIdx.x * 977 + blockIdx.x * 131071ull);
    if constexpr (MODE == 0) {
        for (int it = 0; it < iters; ++it) dft_regs<16, false>(x);
    } else if constexpr (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
            PL p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = pl_from(x[i]);
            pl_dft16(p);
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = pl_canon(p[i]);
        }
    } else {
        PL p[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = pl_from(x[i]);
        for (int it = 0; it < iters; ++it) {
            pl_dft16(p);
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = pl_norm(p[i]);            // keeps the limbs bounded across iterations (2 bits per transform otherwise)
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = pl_canon(p[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16 + i] = x[i];
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const size_t n = (size_t)cus * 8 * 256 * 16;
    u64 *o0, *o1, *o2;
    hipMalloc(&o0, n * 8); hipMalloc(&o1, n * 8); hipMalloc(&o2, n * 8);
    // correctness: one and three transforms
    for (int iters : {1, 3}) {
        hipLaunchKernelGGL(kern<0>, dim3(cus), dim3(256), 0, 0, o0, 12345ULL, iters);
        hipLaunchKernelGGL(kern<1>, dim3(cus), dim3(256), 0, 0, o1, 12345ULL, iters);
        hipLaunchKernelGGL(kern<2>, dim3(cus), dim3(256), 0, 0, o2, 12345ULL, iters);
        hipDeviceSynchronize();
        std::vector<u64> a((size_t)cus * 256 * 16), b(a.size()), c(a.size());
        hipMemcpy(a.data(), o0, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), o2, a.size() * 8, hipMemcpyDeviceToHost);
        size_t bad1 = 0, bad2 = 0;
        for (size_t i = 0; i < a.size(); ++i) { bad1 += a[i] != b[i]; bad2 += a[i] != c[i]; }
        printf("%d transform(s): phi-limb form vs canonical form: %zu / %zu mismatches of %zu (conversions every transform / outside)\n", iters, bad1, bad2, a.size());
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"canonical (modp.cuh)", "phi limbs, canonical in/out per transform", "phi limbs, butterflies + normalisation only"};
    typedef void (*kern_t)(u64 *, u64, int);
    kern_t ks[] = {kern<0>, kern<1>, kern<2>};
    for (int occ : {2, 4, 8})
        for (int m = 0; m < 3; ++m) {
            const int blocks = cus * occ;
            hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), 0, 0, o0, 999ULL, ITERS);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(ks[m], dim3(blocks), dim3(256), 0, 0, o0, 999ULL, ITERS);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double dfts = 3.0 * blocks * 256.0 * ITERS;
            printf("%d workgroups of 256 per CU, %-48s: %8.2f G points/s\n", occ, names[m], dfts * 16 / (ms * 1e-3) / 1e9);
        }
    return 0;
}
