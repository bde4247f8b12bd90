// tools/ubench_onewg.hip -- where the time of the 32K-point one-workgroup half-transform goes (ntt_onewg.cuh, persistent
// form): per-phase s_memtime stamps of one wave, and ablations of the memory-side pieces (store pattern, table loads,
// sample prefetch; 16 = every workgroup runs the parity-0 code, to see what the two parities' code paths cost in the shared instruction cache).  Timing only: results are not checked here (tools/ow_ab.cpp and the parity tests do that).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_onewg.hip -o tools/ubench_onewg
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../cuhe_amd/csrc/host_math.hpp"
#include "../cuhe_amd/csrc/ntt_onewg.cuh"

using namespace cuhe;
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { kPhases = 10 };
// variant bit 32 (the un-overlapped exchange time): the 32-point stage that FOLLOWS an exchange is done as two 16-point
// transforms and 16 combining butterflies (the same 80 butterflies), and the waves that do not write in the second round of the exchange
// run their first 16-point transform -- on the values the first round delivered -- while the other half of the workgroup writes; the
// writers of round 1 run theirs after the exchange.  TIMING ONLY: the real kernel would have to deliver the even-indexed values in round 0
// (a permutation of which wave holds which row), the arithmetic here is the same amount of the same instructions on the wrong pairs.
__device__ __forceinline__ void half16(u64 (&v)[32], int h) {
    u64 (&sub)[16] = *reinterpret_cast<u64(*)[16]>(&v[16 * h]);
    dft_regs<16, false>(sub);
}
template <int J>
struct Combine16 {
    static __device__ __forceinline__ void run(u64 (&v)[32]) {
        const u64 a = v[J], b = v[16 + J];
        v[J] = addp(a, b);
        v[16 + J] = sub_shlp<6 * J>(a, b);
        if constexpr (J + 1 < 16) Combine16<J + 1>::run(v);
    }
};
// variant bits: 1 = contiguous stores ([even | odd] halves instead of X[2k + h]); 2 = no stores (kept alive by an impossible
// condition); 4 = no stage-1 table loads (the sample itself stands in); 8 = samples by plain global loads (no LDS-DMA prefetch)
template <int H, int variant>
__device__ __forceinline__ void loop(u64 *dst_, const u32 *src, const u64 *TW1, u64 *buf, const u64 *tw2, int nbatch,
                                     unsigned long long *stamps, int hs) {
    constexpr int T = 1024, Lh = 32768;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int nitems = 2 * nbatch;
    const u32 lds_base = (u32)(uintptr_t)buf;
    auto fetch = [&](int i) {
        const int b = (i >> 4) * 8 + (i & 7);
        const char *row = (const char *)(src + (long)b * Lh) + lane * 16;
#pragma unroll
        for (int p = 0; p < 8; ++p) { const u32 off = (u32)(wave * 8 + p) * 1024u; glds16(row + off, lds_base + off); }
    };
    unsigned long long acc[kPhases] = {0};
    constexpr bool dma = !(variant & 8);
    if (dma && (int)blockIdx.x < nitems) fetch(blockIdx.x);
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int batch = (item >> 4) * 8 + (item & 7);
        unsigned long long c0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned long long c1 = __builtin_readcyclecounter(); acc[0] += c1 - c0;          // drain of stores + DMA, barrier
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        u64 *lb = buf + opaque;
        u64 x[32], y[32], z[32];
        if constexpr (dma) {
            const u32 *in = (const u32 *)lb + t;
#pragma unroll
            for (int a = 0; a < 32; ++a) x[a] = in[a * T];
        } else {
            const u32 *in = src + (long)batch * Lh + t + opaque;
#pragma unroll
            for (int a = 0; a < 32; ++a) x[a] = in[a * T];
        }
        __syncthreads();
        unsigned long long c2 = __builtin_readcyclecounter(); acc[1] += c2 - c1;          // samples -> registers, barrier
        if constexpr (H) HalfShift<0>::run(x);
        const u64 *t1 = TW1 + (long)H * Lh + t + opaque;
        if constexpr (variant & 4) {
            dft_regs<32, false>(x);
#pragma unroll
            for (int ka = 1; ka < 32; ++ka) x[bitrev<32>(ka)] = mulp(x[bitrev<32>(ka)], x[0] | 1);
        } else ow_dft_twiddle<1024>(x, t1, H != 0);
        unsigned long long c3 = __builtin_readcyclecounter(); acc[2] += c3 - c2;          // shifts, 32-point transform, table products
        const int lo = t & 31, hi = t >> 5;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if ((hi >> 4) == hh) {
                u64 *wr = lb + lo * 545 + (hi & 15);
#pragma unroll
                for (int ka = 0; ka < 32; ++ka) wr[ka * 17] = x[bitrev<32>(ka)];
            } else if constexpr ((variant & 32) != 0) {
                if (hh == 1) half16(y, 0);                                               // round-0 writers: compute while round 1 is written
            }
            __syncthreads();
            const u64 *rd = lb + hi * 545 + lo * 17;
#pragma unroll
            for (int bl = 0; bl < 16; ++bl) y[16 * hh + bl] = rd[bl];
            __syncthreads();
        }
        unsigned long long c4 = __builtin_readcyclecounter(); acc[3] += c4 - c3;          // exchange 1 (4 barriers)
        if constexpr ((variant & 32) != 0) {
            if ((hi >> 4) == 1) half16(y, 0);
            half16(y, 1);
            Combine16<0>::run(y);
        } else dft_regs<32, false>(y);
#pragma unroll
        for (int kb = 1; kb < 32; ++kb) y[bitrev<32>(kb)] = mulp(y[bitrev<32>(kb)], tw2[32 * kb + hi + opaque]);
        unsigned long long c5 = __builtin_readcyclecounter(); acc[4] += c5 - c4;          // stage 2
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if ((hi >> 4) == hh) {
                u64 *wr = lb + lo * 17 + (hi & 15);
#pragma unroll
                for (int kb = 0; kb < 32; ++kb) wr[kb * 544] = y[bitrev<32>(kb)];
            } else if constexpr ((variant & 32) != 0) {
                if (hh == 1) half16(z, 0);
            }
            __syncthreads();
            const u64 *rd = lb + hi * 544 + lo * 17;
#pragma unroll
            for (int cl = 0; cl < 16; ++cl) z[16 * hh + cl] = rd[cl];
            __syncthreads();
        }
        unsigned long long c6 = __builtin_readcyclecounter(); acc[5] += c6 - c5;          // exchange 2 (4 barriers)
        if (dma && item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
        if constexpr ((variant & 32) != 0) {
            if ((hi >> 4) == 1) half16(z, 0);
            half16(z, 1);
            Combine16<0>::run(z);
        } else dft_regs<32, false>(z);
        unsigned long long c7 = __builtin_readcyclecounter(); acc[6] += c7 - c6;          // prefetch issue + stage 3
        if constexpr (variant & 2) {
            if (z[3] == 0x123456789abcdef0ull) dst_[t] = z[5];
        } else if constexpr (variant & 1) {
            u64 *dst = dst_ + (long)batch * 65536 + (long)hs * Lh;
#pragma unroll
            for (int kc = 0; kc < 32; ++kc) dst[t + T * kc] = z[bitrev<32>(kc)];
        } else {
            u64 *dst = dst_ + (long)batch * 65536 + hs;
#pragma unroll
            for (int kc = 0; kc < 32; ++kc) dst[2L * (t + T * kc)] = z[bitrev<32>(kc)];
        }
        unsigned long long c8 = __builtin_readcyclecounter(); acc[7] += c8 - c7;          // store issue
        acc[8] += 1;
    }
    if (lane == 0 && (wave == 0 || wave == 15) && stamps) {
        unsigned long long *o = stamps + ((long)blockIdx.x * 2 + (wave ? 1 : 0)) * kPhases;
        for (int i = 0; i < kPhases; ++i) o[i] = acc[i];
    }
}
template <int variant>
__global__ __launch_bounds__(1024, 4)
void k_stream(u64 *dst, const u32 *src, const u64 *TW1, const u64 *TW2, int nbatch, unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 *buf = lds, *tw2 = lds + OwGeom<32>::XW;
    tw2[threadIdx.x] = TW2[threadIdx.x];
    const int hs = (blockIdx.x >> 3) & 1;                // where the results go: always the workgroup's own parity (same memory traffic in every variant)
    if (hs && !(variant & 16)) loop<1, variant>(dst, src, TW1, buf, tw2, nbatch, stamps, hs);
    else loop<0, variant>(dst, src, TW1, buf, tw2, nbatch, stamps, hs);
}

int main(int argc, char **argv) {
    const int nbatch = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 5;
    const int Lh = 32768, T = 1024;
    const u64 W = host::G;
    std::vector<u64> r(65536);
    r[0] = 1;
    for (int i = 1; i < 65536; ++i) r[i] = host::mulP(r[i - 1], W);
    std::vector<u64> fh(2 * (size_t)Lh), t2(T);
    for (int ka = 0; ka < 32; ++ka)
        for (int m = 0; m < T; ++m) {
            fh[(size_t)ka * T + m] = r[(2L * m * ka) % 65536];
            fh[Lh + (size_t)ka * T + m] = r[((long)m * (2 * ka + 1)) % 65536];
        }
    for (int kb = 0; kb < 32; ++kb) for (int c = 0; c < 32; ++c) t2[kb * 32 + c] = r[(64L * c * kb) % 65536];
    std::vector<u32> hx((size_t)nbatch * Lh);
    unsigned long long s = 1;
    for (auto &v : hx) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (u32)(s >> 32); }
    u64 *dTW1, *dTW2, *dst; u32 *dsrc; unsigned long long *dst_amps;
    HK(hipMalloc(&dTW1, fh.size() * 8)); HK(hipMalloc(&dTW2, t2.size() * 8)); HK(hipMalloc(&dsrc, hx.size() * 4));
    HK(hipMalloc(&dst, (size_t)nbatch * 65536 * 8)); HK(hipMalloc(&dst_amps, 256 * 2 * kPhases * 8));
    HK(hipMemcpy(dTW1, fh.data(), fh.size() * 8, hipMemcpyHostToDevice)); HK(hipMemcpy(dTW2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(dsrc, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = OwGeom<32>::bytes;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"drain+barrier", "samples->regs", "stage1 dft+tw", "exchange 1", "stage 2", "exchange 2", "prefetch+stage3", "store issue"};
    typedef void (*kern_t)(u64 *, const u32 *, const u64 *, const u64 *, int, unsigned long long *);
    struct V { int variant; kern_t k; } vs[] = {{1, k_stream<1>}, {33, k_stream<33>}, {2, k_stream<2>}, {34, k_stream<34>}, {1, k_stream<1>}, {33, k_stream<33>}, {2, k_stream<2>}, {34, k_stream<34>}};
    if (argc > 3) { V all[] = {{0, k_stream<0>}, {1, k_stream<1>}, {2, k_stream<2>}, {16, k_stream<16>}, {17, k_stream<17>}, {0, k_stream<0>}, {16, k_stream<16>}}; (void)all; }
    for (auto &v : vs) {
        const int variant = v.variant;
        HK(hipFuncSetAttribute((const void *)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(v.k, dim3(256), dim3(1024), lds, 0, dst, dsrc, dTW1, dTW2, nbatch, (unsigned long long *)nullptr);
        HK(hipDeviceSynchronize());
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(v.k, dim3(256), dim3(1024), lds, 0, dst, dsrc, dTW1, dTW2, nbatch, i == iters - 1 ? dst_amps : nullptr);
        hipEventRecord(e1, 0);
        HK(hipEventSynchronize(e1));
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double per = ms / iters * 1e-3 / nbatch;
        printf("variant %2d [%s%s%s%s]: %.4f ms per %d transforms, %.3f M/s, frac %.4f\n", variant, variant & 1 ? "contiguous-stores " : "", variant & 2 ? "no-stores " : "",
               variant & 4 ? "no-table-loads " : "", variant & 32 ? "split-stage-overlap " : variant & 8 ? "plain-loads " : variant & 16 ? "one-code-path " : "", ms / iters, nbatch, 1e-6 / per, 655360.0 / per / 8e12);
        std::vector<unsigned long long> st(256 * 2 * kPhases);
        HK(hipMemcpy(st.data(), dst_amps, st.size() * 8, hipMemcpyDeviceToHost));
        for (int blk : {0, 8, 100}) for (int wv = 0; wv < 2; ++wv) {
            const unsigned long long *o = &st[((size_t)blk * 2 + wv) * kPhases];
            const double n = (double)o[8];
            double tot = 0; for (int i = 0; i < 8; ++i) tot += o[i];
            printf("   block %3d wave %2d: items %.0f, cycles per item %.0f:", blk, wv ? 15 : 0, n, tot / n);
            for (int i = 0; i < 8; ++i) printf(" %s %.0f |", names[i], o[i] / n);
            printf("\n");
        }
    }
    return 0;
}
