// ubench_modp.hip -- throughput of the mod-P field primitives as compiled for gfx950,
// plus candidate re-implementations.  Each kernel applies OP to 16 independent
// accumulator pairs per thread, ITERS times.  Prints lane-ops/s per primitive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../cuhe_amd/csrc/modp.cuh"
using namespace cuhe;

#define ITERS 512
#define NACC 16

// ---- candidate: carry-chain add/sub without v_cndmask / v_cmp_u64
__device__ __forceinline__ u64 addp_cc(u64 a, u64 b) {
    // r = a - (P - b), +P on borrow
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 n0, n1, d0, d1, m;
    asm volatile(
        "v_sub_co_u32 %0, vcc, 1, %5\n\ts_nop 1\n\t"
        "v_subb_co_u32 %1, vcc, -1, %6, vcc\n\t"
        "v_sub_co_u32 %2, vcc, %7, %0\n\ts_nop 1\n\t"
        "v_subb_co_u32 %3, vcc, %8, %1, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32 %4, s[2:3], %2, %2, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %3, vcc, %4, %3, vcc"
        : "=&v"(n0), "=&v"(n1), "=&v"(d0), "=&v"(d1), "=&v"(m)
        : "v"(b0), "v"(b1), "v"(a0), "v"(a1)
        : "vcc", "s2", "s3");
    return ((u64)d1 << 32) | d0;
}
__device__ __forceinline__ u64 subp_cc(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 d0, d1, m;
    asm volatile(
        "v_sub_co_u32 %0, vcc, %5, %3\n\ts_nop 1\n\t"
        "v_subb_co_u32 %1, vcc, %6, %4, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32 %2, s[2:3], %0, %0, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %0, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %2, %1, vcc"
        : "=&v"(d0), "=&v"(d1), "=&v"(m)
        : "v"(b0), "v"(b1), "v"(a0), "v"(a1)
        : "vcc", "s2", "s3");
    return ((u64)d1 << 32) | d0;
}
// ---- candidate: select-free C versions (let the compiler pick)
__device__ __forceinline__ u64 subp_m(u64 a, u64 b) {
    u64 d = a - b;
    u64 m = (u64)0 - (u64)(a < b);           // all ones on borrow
    return d - (m & kEps);
}
__device__ __forceinline__ u64 addp_m(u64 a, u64 b) { return subp_m(a, kP - b); }

// ---- candidate shift: one correction for (carry of lo + mid*eps) and (result >= P), which exclude each other
template <int K>
__device__ __forceinline__ u64 shlp_v2(u64 x) {
    static_assert(K > 0 && K < 32, "");
    u64 lo = x << K;
    u32 mid = (u32)(x >> (64 - K));
    u64 r = (u64)mid * 0xffffffffu + lo;       // one v_mad_u64_u32
    u64 t = r + kEps;
    return ((r < lo) | (t < r)) ? t : r;
}
__device__ __forceinline__ u64 shl32_v2(u64 y) {    // y * 2^32 = y0*phi + y1*(phi-1)
    u32 y0 = (u32)y, y1 = (u32)(y >> 32);
    u64 lo = (u64)y0 << 32;
    u64 r = (u64)y1 * 0xffffffffu + lo;
    u64 t = r + kEps;
    return ((r < lo) | (t < r)) ? t : r;
}
template <int K>
__device__ __forceinline__ u64 shlp_v2_mid(u64 x) { return shl32_v2(shlp_v2<K - 32>(x)); }
// mul with merged corrections
__device__ __forceinline__ u64 mulp_v2(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 t = (u64)a0 * b0;
    u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v = (u64)a1 * b0 + (u32)u;
    u64 w = (u64)a1 * b1 + (u >> 32) + (v >> 32);
    u64 lo = (v << 32) | (u32)t;
    u32 hh = (u32)(w >> 32), hl = (u32)w;
    // lo + hl*eps - hh  ==  lo + hl*eps + (P - hh) - P ; P - hh = (eps - hh... ) keep it simple: two steps
    u64 r = (u64)hl * 0xffffffffu + lo;
    u64 t1 = r + kEps;
    r = ((r < lo) | (t1 < r)) ? t1 : r;        // canonical lo + hl*eps
    u64 d = r - hh;
    return (r < hh) ? d - kEps : d;
}
enum { OP_SHL7_V2 = 100, OP_SHL45_V2, OP_MUL_V2 };
enum { OP_ADD, OP_SUB, OP_MUL, OP_SHL7, OP_SHL45, OP_SHL72, OP_ADD_CC, OP_SUB_CC, OP_ADD_M, OP_SUB_M, OP_BFLY, OP_BFLY_CC, OP_MODSMALL };

template <int OP>
__global__ __launch_bounds__(256) void kern(u64 *out, u64 seed) {
    u64 x[NACC], y[NACC];
    for (int i = 0; i < NACC; ++i) { x[i] = canon(seed * (2 * i + 3) + threadIdx.x); y[i] = canon(seed * (2 * i + 5) + 7 * threadIdx.x); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == OP_ADD) x[i] = addp(x[i], y[i]);
            else if (OP == OP_SUB) x[i] = subp(x[i], y[i]);
            else if (OP == OP_MUL) x[i] = mulp(x[i], y[i]);
            else if (OP == OP_SHL7) x[i] = shlp<7>(x[i]);
            else if (OP == OP_SHL45) x[i] = shlp<45>(x[i]);
            else if (OP == OP_SHL72) x[i] = shlp<72>(x[i]);
            else if (OP == OP_ADD_CC) x[i] = addp_cc(x[i], y[i]);
            else if (OP == OP_SUB_CC) x[i] = subp_cc(x[i], y[i]);
            else if (OP == OP_ADD_M) x[i] = addp_m(x[i], y[i]);
            else if (OP == OP_SUB_M) x[i] = subp_m(x[i], y[i]);
            else if (OP == OP_BFLY) { u64 u = x[i], v = y[i]; x[i] = addp(u, v); y[i] = subp(u, v); }
            else if (OP == OP_BFLY_CC) { u64 u = x[i], v = y[i]; x[i] = addp_cc(u, v); y[i] = subp_cc(u, v); }
            else if (OP == OP_SHL7_V2) x[i] = shlp_v2<7>(x[i]);
            else if (OP == OP_SHL45_V2) x[i] = shlp_v2_mid<45>(x[i]);
            else if (OP == OP_MUL_V2) x[i] = mulp_v2(x[i], y[i]);
            else if (OP == OP_MODSMALL) x[i] = x[i] * 3 + mod_small(x[i], 16777213u, 0x10000030000ULL);
        }
    }
    u64 acc = 0;
    for (int i = 0; i < NACC; ++i) acc ^= x[i] ^ y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run(const char *name, u64 *out, int cus, int blocks_per_cu, double ops_per_call) {
    int blocks = cus * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345ULL);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern<OP>, dim3(blocks), dim3(256), 0, 0, out, 12345ULL);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double calls = 3.0 * blocks * 256.0 * ITERS * NACC * ops_per_call;
    double rate = calls / (ms * 1e-3);
    printf("%-12s occ %d blk/CU: %7.3f T field-ops/s  => %.2f clk/op/lane-slot (128 lanes/clk/CU @2.4GHz)\n", name, blocks_per_cu,
           rate / 1e12, 128.0 * 2.4e9 * cus / rate);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    u64 *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    for (int occ : {2}) {
        run<OP_ADD>("addp", out, cus, occ, 1);
        run<OP_SUB>("subp", out, cus, occ, 1);
        run<OP_ADD_CC>("addp_cc", out, cus, occ, 1);
        run<OP_SUB_CC>("subp_cc", out, cus, occ, 1);
        run<OP_ADD_M>("addp_m", out, cus, occ, 1);
        run<OP_SUB_M>("subp_m", out, cus, occ, 1);
        run<OP_BFLY>("bfly", out, cus, occ, 2);
        run<OP_BFLY_CC>("bfly_cc", out, cus, occ, 2);
        run<OP_MUL>("mulp", out, cus, occ, 1);
        run<OP_SHL7>("shlp<7>", out, cus, occ, 1);
        run<OP_SHL45>("shlp<45>", out, cus, occ, 1);
        run<OP_SHL72>("shlp<72>", out, cus, occ, 1);
        run<OP_SHL7_V2>("shlp_v2<7>", out, cus, occ, 1);
        run<OP_SHL45_V2>("shlp_v2<45>", out, cus, occ, 1);
        run<OP_MUL_V2>("mulp_v2", out, cus, occ, 1);
        run<OP_MODSMALL>("mod_small", out, cus, occ, 1);
    }
    return 0;
}
