// ntt_onewg.hpp -- host-side entry of the one-workgroup transforms (ntt_onewg.cuh).  The kernels are instantiated in
// their own translation units (ntt_onewg_inst.hip, compiled once per sub-transform size) so that the sizes build in
// parallel; cuhe_transforms.hip calls them through ow_launch.
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_kernels.cuh"

namespace cuhe {

struct OwArgs {
    void *dst; const void *src;
    const u64 *TW1, *TW2;                 // stage-1 table (HALF: both parities, u64[2][Lh]) and stage-2 table u64[Lh/32]
    long src_stride, dst_stride; int nbatch, nstore;
    WindowArgs wa; const u64 *tw;         // kSrcWindow geometry; twist table (kSrcU32Twist) or second operand rows (kSrcU64NegMul)
    const u32 *primes; const u64 *pinv; int prime0, np_mod;
    const u32 *aux; long aux_stride; FoldGeom fg; const u64 *xtab;
    u64 c128 = 0; int i4neg = 0;          // split negacyclic forward rows: psi^T and the sign of psi^Lh = +-2^48 (ntt_onewg.cuh)
    const RowRebase *rb = nullptr;        // rows in separate blocks (ntt_kernels.cuh); ow_launch_* only, not the persistent forms
};
// hipErrorInvalidValue: this (sub-transform size, source, epilogue, half) combination is not instantiated
hipError_t ow_launch_12(int mode, int out, bool half, const OwArgs &a, hipStream_t st);        // 4K-point halves of the zero-padded 8K-point transform only
hipError_t ow_launch_13(int mode, int out, bool half, const OwArgs &a, hipStream_t st);
hipError_t ow_launch_14(int mode, int out, bool half, const OwArgs &a, hipStream_t st);
hipError_t ow_launch_15(int mode, int out, bool half, const OwArgs &a, hipStream_t st);
// persistent form for the halves of rows of 32K / 64K points (sub-transforms of 2^14 / 2^15 points): kSrcU32Ext
// (zero-padded forward, a.TW1 = the parity tables) or, 64K-point rows only, kSrcU32Twist (negacyclic forward of full rows,
// a.TW1 = the twisted tables, c128 = psi^1024, i4neg: psi^32768 = -2^48); `grid`: resident workgroups, a multiple of 16;
// pair_cnt: grid / 2 <= kOwPairCounters counters for the rendezvous of the two halves of a row (or null)
constexpr int kOwPairCounters = 512;     // + 1 slot behind them: workgroups that gave a rendezvous up (never reset by a launch)
hipError_t ow_launch_stream_14(int mode, int out, const OwArgs &a, int grid, unsigned *pair_cnt, u64 c128, int i4neg, hipStream_t st);
hipError_t ow_launch_stream_15(int mode, int out, const OwArgs &a, int grid, unsigned *pair_cnt, u64 c128, int i4neg, hipStream_t st);
// every kernel of one size made ready on the current device (code object, functions, LDS attributes): at initialisation
hipError_t ow_prewarm_12(); hipError_t ow_prewarm_13(); hipError_t ow_prewarm_14(); hipError_t ow_prewarm_15();
bool ow_supported(int mode, int out, bool half);
// half = true with a full-length source: the SPLIT form (two half-length transforms per row), 32K / 64K-point rows
bool ow_split_supported(int mode, int out);

}  // namespace cuhe
