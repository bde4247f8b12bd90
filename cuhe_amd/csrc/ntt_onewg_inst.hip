// ntt_onewg_inst.hip -- instantiations + launcher of the one-workgroup transforms for ONE sub-transform size
// (-DCUHE_OW_LGH=12|13|14|15; cuhe_amd/build.py compiles the sizes in parallel; 12 = the 4K-point halves of the
// zero-padded 8K-point transform only).
#include "ntt_onewg.hpp"
#include "ntt_onewg.cuh"

#include <atomic>
#include <mutex>

#ifndef CUHE_OW_LGH
#error "compile with -DCUHE_OW_LGH=12, 13, 14 or 15"
#endif

namespace cuhe {
namespace {

constexpr int kLgh = CUHE_OW_LGH;
const RowRebase kNoRebase{};                          // per = 0: rows of one array
using Geo = OwGeom<(1 << kLgh) / 1024>;

// the large-LDS attribute once per (instantiation, device); host threads may race to be first.  Setting it is also what makes the
// runtime load this translation unit's code object and build the kernel's function for the device (HIP defers both to the first use:
// 1-3 ms the first time a size is touched) -- ow_prewarm_* below does it for every instantiation at initialisation.
template <int MODE, int OUT, bool HALF>
hipError_t attr_once() {
    auto kern = ntt_onewg<kLgh, MODE, OUT, HALF>;
    static std::mutex mu; static std::atomic<uint64_t> done{0};
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (cur & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        std::lock_guard<std::mutex> lk(mu);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo::bytes);
            if (e != hipSuccess) return e;
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    return hipSuccess;
}
template <int MODE, int OUT, bool HALF>
hipError_t launch(const OwArgs &a, hipStream_t st) {
    auto kern = ntt_onewg<kLgh, MODE, OUT, HALF>;
    hipError_t e = attr_once<MODE, OUT, HALF>();
    if (e != hipSuccess) return e;
    const int nb8 = (a.nbatch + 7) & ~7;
    const int grid = HALF ? 2 * nb8 : a.nbatch;
#ifdef CUHE_OW_NO_REBASE_ARG
    if (a.rb) return hipErrorInvalidValue;            // (A/B build: no list launches)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Geo::T), Geo::bytes, st, a.dst, a.src, a.TW1, a.TW2, a.src_stride, a.dst_stride, a.nbatch,
                       a.nstore, a.wa, a.tw, a.primes, a.pinv, a.prime0, a.np_mod, a.aux, a.aux_stride, a.fg, a.xtab, StreamTwistArgs{a.c128, a.i4neg});
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Geo::T), Geo::bytes, st, a.dst, a.src, a.TW1, a.TW2, a.src_stride, a.dst_stride, a.nbatch,
                       a.nstore, a.wa, a.tw, a.primes, a.pinv, a.prime0, a.np_mod, a.aux, a.aux_stride, a.fg, a.xtab, StreamTwistArgs{a.c128, a.i4neg},
                       a.rb ? *a.rb : kNoRebase);
#endif
    return hipGetLastError();
}

}  // namespace

#define OW_CONCAT2(a, b) a##b
#define OW_CONCAT(a, b) OW_CONCAT2(a, b)
// every instantiation of this size: X(source, store, half)
//   the zero-padded forward transform of 2^(LGH+1) points, one parity per workgroup
#define OW_CASES_EXT(X) X(kSrcU32Ext, kOutU64, true) X(kSrcU32Ext, kOutU64Mul, true) X(kSrcWindow, kOutU64, true)
#if CUHE_OW_LGH >= 13
//   full-length transforms of 2^LGH points: negacyclic forward, inverses with their store epilogues
#define OW_CASES_FULL(X) X(kSrcU32Twist, kOutU64, false) X(kSrcU32Twist, kOutU64Mul, false) X(kSrcU64Neg, kOutModP, false) \
    X(kSrcU64Neg, kOutModPFoldXn1, false) X(kSrcU64Neg, kOutModPRevQ, false) X(kSrcU64Neg, kOutFoldFinal, false) X(kSrcU64Neg, kOutModPNc, false) \
    X(kSrcU64NegMul, kOutModP, false) X(kSrcU64NegMul, kOutModPFoldXn1, false) X(kSrcU64NegMul, kOutModPNc, false)
#else
#define OW_CASES_FULL(X)
#endif
#if CUHE_OW_LGH >= 14
//   SPLIT rows of 2^(LGH+1) points (negacyclic forward / inverse), one parity per workgroup
#define OW_CASES_SPLIT(X) X(kSrcU32Twist, kOutU64, true) X(kSrcU32Twist, kOutU64Mul, true) X(kSrcU64Neg, kOutModPNc, true) X(kSrcU64NegMul, kOutModPNc, true)
#else
#define OW_CASES_SPLIT(X)
#endif
#define OW_CASES(X) OW_CASES_EXT(X) OW_CASES_FULL(X) OW_CASES_SPLIT(X)
#define OW_CASE(MODE, OUT, HALF) if (mode == MODE && out == OUT && half == HALF) return launch<MODE, OUT, HALF>(a, st);
hipError_t OW_CONCAT(ow_launch_, CUHE_OW_LGH)(int mode, int out, bool half, const OwArgs &a, hipStream_t st) {
    OW_CASES(OW_CASE)
    return hipErrorInvalidValue;
}
#define OW_WARM(MODE, OUT, HALF) if ((e = attr_once<MODE, OUT, HALF>()) != hipSuccess) return e;
#if CUHE_OW_LGH >= 14
// persistent form of the halves of a row of 2^(LGH+1) points (ntt_onewg_stream): `grid` workgroups (a multiple of 16, every one
// resident: 1 / 2 per CU at 32K / 16K points) walk over the 2 * batch halves.  mode kSrcU32Ext: the zero-padded forward
// transform (a.TW1 = u64[2][Lh], the parity tables of the half mode); kSrcU32Twist: the negacyclic forward transform of
// full rows (a.TW1 = the twisted tables, c128 / i4neg the two constants of the twist).
template <int SRC, int OUT>
static hipError_t attr_once_stream() {
    auto kern = ntt_onewg_stream<kLgh, SRC, OUT>;
    static std::mutex mu; static std::atomic<uint64_t> done{0};
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (cur & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        std::lock_guard<std::mutex> lk(mu);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo::bytes_stream);
            if (e != hipSuccess) return e;
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    return hipSuccess;
}
template <int SRC, int OUT>
static hipError_t launch_stream(const OwArgs &a, int grid, unsigned *pair_cnt, StreamTwistArgs ta, hipStream_t st) {
    auto kern = ntt_onewg_stream<kLgh, SRC, OUT>;
    hipError_t e = attr_once_stream<SRC, OUT>();
    if (e != hipSuccess) return e;
    if (pair_cnt) {                                   // (grid / 2) rendezvous counters of the row pairs, zero at launch
        e = hipMemsetAsync(pair_cnt, 0, (size_t)(grid / 2) * sizeof(unsigned), st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Geo::T), Geo::bytes_stream, st, a.dst, (const u32 *)a.src, a.TW1, a.TW2, a.src_stride, a.dst_stride, a.nbatch,
                       a.xtab, a.prime0, a.np_mod, pair_cnt, ta);
    return hipGetLastError();
}
hipError_t OW_CONCAT(ow_launch_stream_, CUHE_OW_LGH)(int mode, int out, const OwArgs &a, int grid, unsigned *pair_cnt, u64 c128, int i4neg, hipStream_t st) {
    if (grid < 16 || (grid & 15) || grid / 2 > kOwPairCounters) return hipErrorInvalidValue;
    const StreamTwistArgs ta{c128, i4neg};
    if (mode == kSrcU32Ext && out == kOutU64) return launch_stream<kSrcU32Ext, kOutU64>(a, grid, pair_cnt, ta, st);
    if (mode == kSrcU32Ext && out == kOutU64Mul) return launch_stream<kSrcU32Ext, kOutU64Mul>(a, grid, pair_cnt, ta, st);
#if CUHE_OW_LGH == 15
    if (mode == kSrcU32Twist && out == kOutU64) return launch_stream<kSrcU32Twist, kOutU64>(a, grid, pair_cnt, ta, st);
#endif
    return hipErrorInvalidValue;
}
#endif
// every kernel of this size made ready on the current device (code object loaded, functions built, LDS attributes set): called by
// cuhe_hip_init for the sizes the parameter set can reach, so that the first gate that uses a size does not pay for it -- a homomorphic
// PRINCE block's first uses of the 8K / 16K / 32K-point forms cost it 6 of its 9.6 ms of idle GPU (profiles/r06_prince_first_block.txt)
hipError_t OW_CONCAT(ow_prewarm_, CUHE_OW_LGH)() {
    hipError_t e = hipSuccess;
    OW_CASES(OW_WARM)
#if CUHE_OW_LGH >= 14
    if ((e = attr_once_stream<kSrcU32Ext, kOutU64>()) != hipSuccess) return e;
    if ((e = attr_once_stream<kSrcU32Ext, kOutU64Mul>()) != hipSuccess) return e;
#endif
#if CUHE_OW_LGH == 15
    if ((e = attr_once_stream<kSrcU32Twist, kOutU64>()) != hipSuccess) return e;
#endif
    return e;
}
#if CUHE_OW_LGH == 15
bool ow_supported(int mode, int out, bool half) {
    if (half) return (mode == kSrcU32Ext && (out == kOutU64 || out == kOutU64Mul)) || (mode == kSrcWindow && out == kOutU64);
    if (mode == kSrcU32Twist) return out == kOutU64 || out == kOutU64Mul;
    if (mode == kSrcU64Neg) return out == kOutModP || out == kOutModPFoldXn1 || out == kOutModPRevQ || out == kOutFoldFinal || out == kOutModPNc;
    if (mode == kSrcU64NegMul) return out == kOutModP || out == kOutModPFoldXn1 || out == kOutModPNc;
    return false;
}
bool ow_split_supported(int mode, int out) {
    if (mode == kSrcU32Twist) return out == kOutU64 || out == kOutU64Mul;
    return (mode == kSrcU64Neg || mode == kSrcU64NegMul) && out == kOutModPNc;
}
#endif

}  // namespace cuhe
