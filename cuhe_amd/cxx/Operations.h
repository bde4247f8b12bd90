// Operations.h -- raw-pointer operation drivers, API of cuhe/Operations.h:42-108.
// Each function forwards to the C ABI (include/cuhe_hip.h); a failing status
// prints file:line and exits like the reference's CSC/CCE.
#pragma once
#include <NTL/ZZ.h>
#include <NTL/ZZX.h>
NTL_CLIENT

#ifndef CUHE_STREAM_T
#define CUHE_STREAM_T
typedef void *cudaStream_t;      // hipStream_t travels through the C ABI as void*
#endif

typedef unsigned int uint32;
typedef unsigned long int uint64;

namespace cuHE {

// pre-computation
void initCrt(ZZ *coeffModulus);
void loadIcrtConst(int lvl, int dev, cudaStream_t st = 0);
void initNtt();
void initBarrett(ZZX m);
void getCoeffModuli(ZZ *dst);
uint32 *inttResult(int dev);

// CRT domain
void crt(uint32 *dst, uint32 *src, int logq, int dev, cudaStream_t st = 0);
void icrt(uint32 *dst, uint32 *src, int logq, int dev, cudaStream_t st = 0);
void crtAdd(uint32 *sum, uint32 *x, uint32 *y, int logq, int dev, cudaStream_t st = 0);
void crtAddInt(uint32 *sum, uint32 *x, unsigned a, int logq, int dev, cudaStream_t st = 0);
void crtAddNX1(uint32 *sum, uint32 *x, uint32 *scalar, int logq, int dev, cudaStream_t st = 0);
void crtMulInt(uint32 *prod, uint32 *x, int a, int logq, int dev, cudaStream_t st = 0);
void crtModSwitch(uint32 *dst, uint32 *src, int logq, int dev, cudaStream_t st = 0);

// transforms: one polynomial / all CRT polynomials of a level
void _ntt(uint64 *X, uint32 *x, int dev, cudaStream_t st = 0);
void _nttw(uint64 *X, uint32 *x, int coeffwords, int relinIdx, int dev, cudaStream_t st = 0);
void _intt(uint32 *x, uint64 *X, int crtidx, int dev, cudaStream_t st = 0);
void ntt(uint64 *X, uint32 *x, int logq, int dev, cudaStream_t st = 0);
void nttw(uint64 *X, uint32 *x, int logq, int dev, cudaStream_t st = 0);
void intt(uint32 *x, uint64 *X, int logq, int dev, cudaStream_t st = 0);
void inttHold(uint64 *X, int logq, int dev, cudaStream_t st = 0);
void inttDoubleDeg(uint32 *x, uint64 *X, int logq, int dev, cudaStream_t st = 0);
void inttMod(uint32 *x, uint64 *X, int logq, int dev, cudaStream_t st = 0);

// NTT domain
void nttMul(uint64 *z, uint64 *y, uint64 *x, int logq, int dev, cudaStream_t st = 0);
void nttMulNX1(uint64 *z, uint64 *x, uint64 *scalar, int logq, int dev, cudaStream_t st = 0);
void nttAdd(uint64 *z, uint64 *y, uint64 *x, int logq, int dev, cudaStream_t st = 0);
void nttAddNX1(uint64 *z, uint64 *x, uint64 *scalar, int logq, int dev, cudaStream_t st = 0);

// reduction modulo the polynomial modulus
void barrett(uint32 *dst, uint32 *src, int lvl, int dev, cudaStream_t st = 0);
void barrett(uint32 *dst, int lvl, int dev, cudaStream_t st = 0);

// kept for source compatibility: the constants they produced are generated
// inside the C-ABI library at initCuHE time, so these are no-ops after init
void genCrtPrimes();
void genCoeffModuli();
void genCrtInvPrimes();
void genIcrtByLevel(int lvl);
void genIcrt();
void setPolyModulus(ZZX m);
void createBarrettTemporySpace();
// "not called externally" in the reference, but public there (cuhe/Operations.h:131-134): the transform scratch and the
// inttResult buffer per device.  Here every host thread owns its scratch on every device: these return the CALLING
// thread's buffers (the arrays are indexed by device and belong to the calling thread as well).
uint64 **ptrNttSwap();
uint32 **ptrNttHold();
uint64 *ptrNttSwap(int dev);
uint32 *ptrNttHold(int dev);

} // namespace cuHE
