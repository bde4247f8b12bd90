// icrt_mfma.hip -- instantiations and launch of the matrix-core inverse CRT (icrt_mfma.cuh), in a translation unit of
// its own so that it builds beside the others.  Called from launch_icrt (cuhe_keyswitch.hip).
#include "cuhe_internal.hpp"

namespace cuhe_impl {

template <int TILES>
static int launch_tiles(u32 *dst, const u32 *src, const DevCtx &D, const IcrtLevel &I, int np, int W, int batch, long src_ct_stride,
                        long dst_ct_stride, hipStream_t st, const IcrtWindows &wo) {
    const Params &q = G_.prm;
    const size_t lds = icrt_mfma_lds_bytes(TILES, I.ksteps);
    static AttrOnce once;
    if (lds > 64 * 1024) CHK(once.set(k_icrt_mfma<TILES>, 160 * 1024));
    // resident workgroups loop over the tiles (32 coefficients each, a wave per tile): the 30 KB of constants are staged once
    const long tiles = (long)((q.modLen + kIcrtMfmaTile - 1) / kIcrtMfmaTile) * batch, wgs = (tiles + 3) / 4;
    // resident workgroups per CU (registers and LDS).  The LDS size follows the level (ksteps), so the answer is kept per
    // K-step count of this instantiation (one cached value served launches with another LDS size); devices of one
    // process are the same part.  Performance only: the kernel walks its tiles with a grid stride.
    static std::atomic<int> occ[64];
    const int slot = I.ksteps >= 0 && I.ksteps < 64 ? I.ksteps : 0;
    int per_cu = occ[slot].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int nb = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_icrt_mfma<TILES>, kIcrtMfmaThreads, lds));
        per_cu = std::max(nb, 1);
        occ[slot].store(per_cu, std::memory_order_relaxed);
    }
    // (A/B: CUHE_ICRT_GRID_MULT workgroups per resident slot -- the copy probe streams 25 % faster with far more workgroups than
    // fit at once than with a resident grid-stride walk, profiles/r06_copy_probe.txt; profiles/r06_icrt_grid_ab.txt says what it does here)
    static const int mult = getenv("CUHE_ICRT_GRID_MULT") ? std::max(1, atoi(getenv("CUHE_ICRT_GRID_MULT"))) : 1;
    const long resident = (long)std::max(D.cus, 1) * per_cu * mult;
    const dim3 grid((unsigned)std::min(wgs, resident)), block(kIcrtMfmaThreads);
    IcrtMfmaTab T{I.dig, I.pc, I.nm, I.tiles, I.ksteps};
    hipLaunchKernelGGL(k_icrt_mfma<TILES>, grid, block, lds, st, dst, src, T, np, W, q.modLen, q.crtLen, src_ct_stride, dst_ct_stride, batch, wo);
    HIPCHK(hipGetLastError());
    return CUHE_OK;
}

bool icrt_mfma_supported(const IcrtLevel &I) { return I.dig != nullptr && I.tiles >= 1 && I.tiles <= 6; }

int launch_icrt_mfma(u32 *dst, const u32 *src, const DevCtx &D, const IcrtLevel &I, int np, int W, int batch, long src_ct_stride, long dst_ct_stride,
                     hipStream_t st, const IcrtWindows &wo) {
    switch (I.tiles) {
    case 1: return launch_tiles<1>(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    case 2: return launch_tiles<2>(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    case 3: return launch_tiles<3>(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    case 4: return launch_tiles<4>(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    case 5: return launch_tiles<5>(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    case 6: return launch_tiles<6>(dst, src, D, I, np, W, batch, src_ct_stride, dst_ct_stride, st, wo);
    }
    return fail(CUHE_EINVAL, "matrix-core ICRT: %d result tiles not instantiated", I.tiles);
}

}  // namespace cuhe_impl
