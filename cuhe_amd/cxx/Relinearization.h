// Relinearization.h -- API of cuhe/Relinearization.h.
#pragma once
#include "Operations.h"

namespace cuHE {
// converts every evaluation key to the NTT domain ONCE; keys stay resident in HBM
void initRelin(ZZX *evalkey);
// dst (NTT domain) = sum_j NTT(window_j(src)) * EK_j for every CRT prime of `lvl`
void relinearization(uint64 *dst, uint32 *src, int lvl, int dev, cudaStream_t st = 0);
// (additions) binary cache of the NTT-domain keys initRelin computes -- format in include/cuhe_hip.h.
// save: after initRelin; load: after initCuHE, instead of initRelin.  load returns false (and prints why on
// stderr) if the file is missing, damaged or was made for other parameters / CRT primes.
void saveRelinearization(const char *path);
bool loadRelinearization(const char *path);
} // namespace cuHE
