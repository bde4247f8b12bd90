// ubench_valu.hip -- VALU issue-rate microbenchmark for the integer instructions the
// mod-P field arithmetic lowers to on gfx950.  Prints lane-ops/s and the rate relative
// to v_add_u32.  Usage: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITERS 2048
#define REP8(x) x x x x x x x x

#define KERNEL(name, BODY, CLOB)                                                      \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned seed) {       \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;   \
        unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;       \
        unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7; \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;  \
        for (int i = 0; i < ITERS; ++i) {                                              \
            REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                              "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7), \
                              "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : : CLOB);) \
        }                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) \
            ^ (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                         \
    }

// each BODY = 8 independent instructions (one per accumulator)
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0", "memory")
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n v_add_co_u32 %4, vcc, %4, %5\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_add_co_u32 %6, vcc, %6, %7\n v_addc_co_u32 %7, vcc, %7, %0, vcc", "vcc")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0", "memory")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %4\n v_mul_hi_u32 %4, %4, %5\n v_mul_hi_u32 %5, %5, %6\n v_mul_hi_u32 %6, %6, %7\n v_mul_hi_u32 %7, %7, %0", "memory")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %0, %1, %8\n v_mad_u64_u32 %9, vcc, %1, %2, %9\n v_mad_u64_u32 %10, vcc, %2, %3, %10\n v_mad_u64_u32 %11, vcc, %3, %4, %11\n v_mad_u64_u32 %12, vcc, %4, %5, %12\n v_mad_u64_u32 %13, vcc, %5, %6, %13\n v_mad_u64_u32 %14, vcc, %6, %7, %14\n v_mad_u64_u32 %15, vcc, %7, %0, %15", "vcc")
KERNEL(k_lshl_b64, "v_lshlrev_b64 %8, 7, %8\n v_lshlrev_b64 %9, 7, %9\n v_lshlrev_b64 %10, 7, %10\n v_lshlrev_b64 %11, 7, %11\n v_lshlrev_b64 %12, 7, %12\n v_lshlrev_b64 %13, 7, %13\n v_lshlrev_b64 %14, 7, %14\n v_lshlrev_b64 %15, 7, %15", "memory")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %12\n v_lshl_add_u64 %12, %12, 0, %13\n v_lshl_add_u64 %13, %13, 0, %14\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %15, %15, 0, %8", "memory")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %4, 7\n v_alignbit_b32 %4, %4, %5, 7\n v_alignbit_b32 %5, %5, %6, 7\n v_alignbit_b32 %6, %6, %7, 7\n v_alignbit_b32 %7, %7, %0, 7", "memory")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc", "vcc")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %3, %3, %4, %5\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %0\n v_mad_u32_u24 %7, %7, %0, %1", "memory")
KERNEL(k_fma_f64, "v_fma_f64 %16, %16, %17, %18\n v_fma_f64 %17, %17, %18, %19\n v_fma_f64 %18, %18, %19, %20\n v_fma_f64 %19, %19, %20, %21\n v_fma_f64 %20, %20, %21, %22\n v_fma_f64 %21, %21, %22, %23\n v_fma_f64 %22, %22, %23, %16\n v_fma_f64 %23, %23, %16, %17", "memory")
KERNEL(k_sub_co, "v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %1, vcc, %1, %2, vcc\n v_sub_co_u32 %2, vcc, %2, %3\n v_subb_co_u32 %3, vcc, %3, %4, vcc\n v_sub_co_u32 %4, vcc, %4, %5\n v_subb_co_u32 %5, vcc, %5, %6, vcc\n v_sub_co_u32 %6, vcc, %6, %7\n v_subb_co_u32 %7, vcc, %7, %0, vcc", "vcc")
KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, %8, %9\n v_cmp_lt_u64 vcc, %9, %10\n v_cmp_lt_u64 vcc, %10, %11\n v_cmp_lt_u64 vcc, %11, %12\n v_cmp_lt_u64 vcc, %12, %13\n v_cmp_lt_u64 vcc, %13, %14\n v_cmp_lt_u64 vcc, %14, %15\n v_cmp_lt_u64 vcc, %15, %8", "vcc")

typedef void (*kern_t)(unsigned *, unsigned);

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %.0f MHz\n", prop.name, cus, prop.clockRate / 1000.0);
    unsigned *out; 
    int blocks = cus * 8;                        // 8 blocks x 4 waves = 8 waves/SIMD
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    struct { const char *name; kern_t k; } tests[] = {
        {"v_add_u32", k_add_u32}, {"v_add_co/addc_co", k_add_co}, {"v_sub_co/subb_co", k_sub_co}, {"v_mul_lo_u32", k_mul_lo},
        {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshlrev_b64", k_lshl_b64},
        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_alignbit_b32", k_alignbit}, {"v_cndmask_b32", k_cndmask},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_fma_f64", k_fma_f64}, {"v_cmp_lt_u64", k_cmp_u64}};
    double base = 0;
    for (auto &t : tests) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1u);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double laneops = 5.0 * blocks * 256.0 * ITERS * 64.0;     // 8 REP x 8 instr per iter
        double rate = laneops / (ms * 1e-3);
        if (!base) base = rate;
        printf("%-20s %8.2f T lane-ops/s   %.3f x v_add_u32   (%.1f lane-ops/clk/CU @2.4GHz)\n", t.name, rate / 1e12, rate / base,
               rate / cus / 2.4e9);
    }
    return 0;
}
