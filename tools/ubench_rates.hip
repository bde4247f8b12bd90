// ubench_rates.hip -- issue cost of the integer VALU instructions on gfx950, in SHADER CYCLES per wave-instruction
// (s_memtime ticks of one wave divided by its instruction count) at 1, 2 and 4 waves per SIMD, plus the wall-clock
// rate of the whole chip and the clock the chip settles at under that instruction (ticks per microsecond).
// Purpose: the cost model behind the field arithmetic of modp.cuh -- which instructions issue in 2 cycles (SIMD-32
// rate), which in 4, what a carry chain through VCC / an SGPR pair costs, and what s_nop padding costs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rates.hip -o tools/ubench_rates && tools/ubench_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ITERS 512
#define REP8(x) x x x x x x x x

#define KERNEL(name, BODY, ...)                                                                                         \
    __global__ __launch_bounds__(256) void name(unsigned *out, unsigned long long *ticks, unsigned seed) {               \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;                              \
        unsigned a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                                  \
        unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;                        \
        unsigned long long s0 = seed, s1 = seed + 1;                                                                     \
        unsigned long long t0 = __builtin_readcyclecounter();                                                            \
        for (int i = 0; i < ITERS; ++i) {                                                                                \
            REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),       \
                              "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7),             \
                              "+s"(s0), "+s"(s1) : : __VA_ARGS__);)                                                             \
        }                                                                                                                \
        unsigned long long t1 = __builtin_readcyclecounter();                                                            \
        if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^                              \
            (unsigned)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) ^ (unsigned)(s0 ^ s1);                                      \
    }

// 8 independent instructions per BODY on the rotating accumulators
#define R8_3(op) op " %0, %0, %1\n " op " %1, %1, %2\n " op " %2, %2, %3\n " op " %3, %3, %4\n " op " %4, %4, %5\n " op " %5, %5, %6\n " op " %6, %6, %7\n " op " %7, %7, %0"
#define R8_4(op) op " %0, %0, %1, %2\n " op " %1, %1, %2, %3\n " op " %2, %2, %3, %4\n " op " %3, %3, %4, %5\n " op " %4, %4, %5, %6\n " op " %5, %5, %6, %7\n " op " %6, %6, %7, %0\n " op " %7, %7, %0, %1"
#define R8_IMM(op, imm) op " %0, " imm ", %0\n " op " %1, " imm ", %1\n " op " %2, " imm ", %2\n " op " %3, " imm ", %3\n " op " %4, " imm ", %4\n " op " %5, " imm ", %5\n " op " %6, " imm ", %6\n " op " %7, " imm ", %7"
#define R8_BFE(op) op " %0, %1, 3, 11\n " op " %1, %2, 3, 11\n " op " %2, %3, 3, 11\n " op " %3, %4, 3, 11\n " op " %4, %5, 3, 11\n " op " %5, %6, 3, 11\n " op " %6, %7, 3, 11\n " op " %7, %0, 3, 11"

KERNEL(k_add_u32, R8_3("v_add_u32"), "memory")
KERNEL(k_sub_u32, R8_3("v_sub_u32"), "memory")
KERNEL(k_and_b32, R8_3("v_and_b32"), "memory")
KERNEL(k_xor_b32, R8_3("v_xor_b32"), "memory")
KERNEL(k_min_u32, R8_3("v_min_u32"), "memory")
KERNEL(k_lshl_b32, R8_IMM("v_lshlrev_b32", "7"), "memory")
KERNEL(k_lshr_b32, R8_IMM("v_lshrrev_b32", "7"), "memory")
KERNEL(k_ashr_i32, R8_IMM("v_ashrrev_i32", "7"), "memory")
KERNEL(k_mov_b32, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0", "memory")
KERNEL(k_bfe_u32, R8_BFE("v_bfe_u32"), "memory")
KERNEL(k_bfe_i32, R8_BFE("v_bfe_i32"), "memory")
KERNEL(k_and_or, R8_4("v_and_or_b32"), "memory")
KERNEL(k_lshl_or, R8_4("v_lshl_or_b32"), "memory")
KERNEL(k_lshl_add_u32, R8_4("v_lshl_add_u32"), "memory")
KERNEL(k_add_lshl, R8_4("v_add_lshl_u32"), "memory")
KERNEL(k_add3, R8_4("v_add3_u32"), "memory")
KERNEL(k_xad, R8_4("v_xad_u32"), "memory")
KERNEL(k_bfi, R8_4("v_bfi_b32"), "memory")
KERNEL(k_perm, R8_4("v_perm_b32"), "memory")
KERNEL(k_alignbit, R8_4("v_alignbit_b32"), "memory")
KERNEL(k_mad_u32_u24, R8_4("v_mad_u32_u24"), "memory")
KERNEL(k_mul_u32_u24, R8_3("v_mul_u32_u24"), "memory")
KERNEL(k_mul_hi_u24, R8_3("v_mul_hi_u32_u24"), "memory")
KERNEL(k_mul_lo, R8_3("v_mul_lo_u32"), "memory")
KERNEL(k_mul_hi, R8_3("v_mul_hi_u32"), "memory")
KERNEL(k_pk_add_u16, R8_3("v_pk_add_u16"), "memory")
KERNEL(k_add_f32, R8_3("v_add_f32"), "memory")
KERNEL(k_fma_f32, R8_4("v_fma_f32"), "memory")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc", "memory")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, %16\n v_cndmask_b32 %1, %1, %2, %16\n v_cndmask_b32 %2, %2, %3, %16\n v_cndmask_b32 %3, %3, %4, %16\n v_cndmask_b32 %4, %4, %5, %16\n v_cndmask_b32 %5, %5, %6, %16\n v_cndmask_b32 %6, %6, %7, %16\n v_cndmask_b32 %7, %7, %0, %16", "memory")
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0", "vcc")
KERNEL(k_cmp_u64, "v_cmp_lt_u64 vcc, %8, %9\n v_cmp_lt_u64 vcc, %9, %10\n v_cmp_lt_u64 vcc, %10, %11\n v_cmp_lt_u64 vcc, %11, %12\n v_cmp_lt_u64 vcc, %12, %13\n v_cmp_lt_u64 vcc, %13, %14\n v_cmp_lt_u64 vcc, %14, %15\n v_cmp_lt_u64 vcc, %15, %8", "vcc")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %0, %1, %8\n v_mad_u64_u32 %9, vcc, %1, %2, %9\n v_mad_u64_u32 %10, vcc, %2, %3, %10\n v_mad_u64_u32 %11, vcc, %3, %4, %11\n v_mad_u64_u32 %12, vcc, %4, %5, %12\n v_mad_u64_u32 %13, vcc, %5, %6, %13\n v_mad_u64_u32 %14, vcc, %6, %7, %14\n v_mad_u64_u32 %15, vcc, %7, %0, %15", "vcc")
KERNEL(k_lshl_b64, "v_lshlrev_b64 %8, 7, %8\n v_lshlrev_b64 %9, 7, %9\n v_lshlrev_b64 %10, 7, %10\n v_lshlrev_b64 %11, 7, %11\n v_lshlrev_b64 %12, 7, %12\n v_lshlrev_b64 %13, 7, %13\n v_lshlrev_b64 %14, 7, %14\n v_lshlrev_b64 %15, 7, %15", "memory")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %12\n v_lshl_add_u64 %12, %12, 0, %13\n v_lshl_add_u64 %13, %13, 0, %14\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %15, %15, 0, %8", "memory")
// carry chains: 4 two-instruction 64-bit adds through VCC, no padding / the compiler's s_nop 1 padding
KERNEL(k_add_co_vcc, "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n v_add_co_u32 %4, vcc, %4, %5\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_add_co_u32 %6, vcc, %6, %7\n v_addc_co_u32 %7, vcc, %7, %0, vcc", "vcc")
KERNEL(k_add_co_vcc_nop, "v_add_co_u32 %0, vcc, %0, %1\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %2, vcc, %2, %3\n s_nop 1\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n v_add_co_u32 %4, vcc, %4, %5\n s_nop 1\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_add_co_u32 %6, vcc, %6, %7\n s_nop 1\n v_addc_co_u32 %7, vcc, %7, %0, vcc", "vcc")
// the same with an independent VALU instruction between producer and consumer of the carry instead of the s_nop
KERNEL(k_add_co_interleaved, "v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, %16, %2, %3\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_addc_co_u32 %3, %16, %3, %4, %16\n v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %6, %16, %6, %7\n v_addc_co_u32 %5, vcc, %5, %6, vcc\n v_addc_co_u32 %7, %16, %7, %0, %16", "vcc")
// VALU with interleaved SALU (s_or_b64 on an unrelated pair): does the scalar instruction cost the wave an issue slot?
KERNEL(k_add_u32_salu, "v_add_u32 %0, %0, %1\n s_or_b64 %17, %17, %17\n v_add_u32 %1, %1, %2\n s_or_b64 %17, %17, %17\n v_add_u32 %2, %2, %3\n s_or_b64 %17, %17, %17\n v_add_u32 %3, %3, %4\n s_or_b64 %17, %17, %17\n v_add_u32 %4, %4, %5\n s_or_b64 %17, %17, %17\n v_add_u32 %5, %5, %6\n s_or_b64 %17, %17, %17\n v_add_u32 %6, %6, %7\n s_or_b64 %17, %17, %17\n v_add_u32 %7, %7, %0\n s_or_b64 %17, %17, %17", "scc")
KERNEL(k_mad64_salu, "v_mad_u64_u32 %8, vcc, %0, %1, %8\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %9, vcc, %1, %2, %9\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %10, vcc, %2, %3, %10\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %11, vcc, %3, %4, %11\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %12, vcc, %4, %5, %12\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %13, vcc, %5, %6, %13\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %14, vcc, %6, %7, %14\n s_or_b64 %17, %17, %17\n v_mad_u64_u32 %15, vcc, %7, %0, %15\n s_or_b64 %17, %17, %17", "vcc", "scc")

// v_cndmask forms: e64 encoding naming vcc explicitly; compare + e32 select pairs (what the compiler emits for `c ? x : y`);
// compare into an SGPR pair + e64 select with the compiler's two wait states between them
KERNEL(k_cnd_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %2, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n v_cndmask_b32_e64 %4, %4, %5, vcc\n v_cndmask_b32_e64 %5, %5, %6, vcc\n v_cndmask_b32_e64 %6, %6, %7, vcc\n v_cndmask_b32_e64 %7, %7, %0, vcc", "memory")
KERNEL(k_cmp_cnd_e32, "v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cmp_lt_u32 vcc, %4, %5\n s_nop 1\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_cmp_lt_u32 vcc, %6, %7\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %1, vcc", "vcc")
KERNEL(k_cmp_cnd_e64, "v_cmp_lt_u32_e64 %16, %0, %1\n s_nop 1\n v_cndmask_b32_e64 %2, %2, %3, %16\n v_cmp_lt_u32_e64 %16, %2, %3\n s_nop 1\n v_cndmask_b32_e64 %4, %4, %5, %16\n v_cmp_lt_u32_e64 %16, %4, %5\n s_nop 1\n v_cndmask_b32_e64 %6, %6, %7, %16\n v_cmp_lt_u32_e64 %16, %6, %7\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, %16", "memory")
// 64-bit compare + two selects (the select form of a field correction), and the same with the two compares interleaved two deep
KERNEL(k_cmp64_cnd2, "v_cmp_lt_u64 vcc, %8, %9\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cmp_lt_u64 vcc, %10, %11\n s_nop 1\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cndmask_b32_e32 %6, %6, %7, vcc", "vcc")
KERNEL(k_or_b32, R8_3("v_or_b32"), "memory")
KERNEL(k_lshl_b32_v, R8_3("v_lshlrev_b32"), "memory")
KERNEL(k_lshr_b32_v, R8_3("v_lshrrev_b32"), "memory")
KERNEL(k_subrev_u32, R8_3("v_subrev_u32"), "memory")
KERNEL(k_max_u32, R8_3("v_max_u32"), "memory")
KERNEL(k_mad_u64_u32_s, "v_mad_u64_u32 %8, %16, %0, %1, %8\n v_mad_u64_u32 %9, %16, %1, %2, %9\n v_mad_u64_u32 %10, %16, %2, %3, %10\n v_mad_u64_u32 %11, %16, %3, %4, %11\n v_mad_u64_u32 %12, %16, %4, %5, %12\n v_mad_u64_u32 %13, %16, %5, %6, %13\n v_mad_u64_u32 %14, %16, %6, %7, %14\n v_mad_u64_u32 %15, %16, %7, %0, %15", "memory")

typedef void (*kern_t)(unsigned *, unsigned long long *, unsigned);

int main(int argc, char **argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, nominal clock %.0f MHz; ITERS %d x 64 wave-instructions per kernel\n", prop.name, cus, prop.clockRate / 1000.0, ITERS);
    printf("%-22s | %-28s | %-28s | %-28s\n", "instruction (8 indep.)", "1 wave/SIMD: cyc/instr  T/s  MHz", "2 waves/SIMD", "4 waves/SIMD");
    struct { const char *name; kern_t k; int per_body; } tests[] = {
        {"v_add_u32", k_add_u32, 8}, {"v_sub_u32", k_sub_u32, 8}, {"v_and_b32", k_and_b32, 8}, {"v_xor_b32", k_xor_b32, 8}, {"v_min_u32", k_min_u32, 8},
        {"v_lshlrev_b32", k_lshl_b32, 8}, {"v_lshrrev_b32", k_lshr_b32, 8}, {"v_ashrrev_i32", k_ashr_i32, 8}, {"v_mov_b32", k_mov_b32, 8},
        {"v_bfe_u32", k_bfe_u32, 8}, {"v_bfe_i32", k_bfe_i32, 8}, {"v_and_or_b32", k_and_or, 8}, {"v_lshl_or_b32", k_lshl_or, 8},
        {"v_lshl_add_u32", k_lshl_add_u32, 8}, {"v_add_lshl_u32", k_add_lshl, 8}, {"v_add3_u32", k_add3, 8}, {"v_xad_u32", k_xad, 8},
        {"v_bfi_b32", k_bfi, 8}, {"v_perm_b32", k_perm, 8}, {"v_alignbit_b32", k_alignbit, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8},
        {"v_mul_u32_u24", k_mul_u32_u24, 8}, {"v_mul_hi_u32_u24", k_mul_hi_u24, 8}, {"v_mul_lo_u32", k_mul_lo, 8}, {"v_mul_hi_u32", k_mul_hi, 8},
        {"v_pk_add_u16", k_pk_add_u16, 8}, {"v_add_f32", k_add_f32, 8}, {"v_fma_f32", k_fma_f32, 8},
        {"v_cndmask_b32 vcc", k_cndmask, 8}, {"v_cndmask_b32 sgpr", k_cndmask_s, 8}, {"v_cmp_lt_u32", k_cmp_u32, 8}, {"v_cmp_lt_u64", k_cmp_u64, 8},
        {"v_mad_u64_u32", k_mad_u64_u32, 8}, {"v_lshlrev_b64", k_lshl_b64, 8}, {"v_lshl_add_u64", k_lshl_add_u64, 8},
        {"add_co;addc_co (vcc)", k_add_co_vcc, 8}, {"add_co;s_nop1;addc_co", k_add_co_vcc_nop, 8}, {"2 chains interleaved", k_add_co_interleaved, 8},
        {"v_cndmask_e64 vcc", k_cnd_e64_vcc, 8}, {"cmp32;nop;cnd_e32 x4", k_cmp_cnd_e32, 8}, {"cmp32;nop;cnd_e64 x4", k_cmp_cnd_e64, 8},
        {"cmp64;nop;2 cnd_e32 x2", k_cmp64_cnd2, 6}, {"v_or_b32", k_or_b32, 8}, {"v_lshlrev_b32 (vgpr)", k_lshl_b32_v, 8}, {"v_lshrrev_b32 (vgpr)", k_lshr_b32_v, 8},
        {"v_subrev_u32", k_subrev_u32, 8}, {"v_max_u32", k_max_u32, 8}, {"v_mad_u64_u32 ->sgpr", k_mad_u64_u32_s, 8},
        {"v_add_u32 + s_or_b64", k_add_u32_salu, 8}, {"v_mad_u64_u32 + s_or", k_mad64_salu, 8}};
    unsigned *out; unsigned long long *ticks;
    const int maxblocks = cus * 4;
    hipMalloc(&out, (size_t)maxblocks * 256 * 4);
    hipMalloc(&ticks, (size_t)maxblocks * 4 * 8);
    std::vector<unsigned long long> h((size_t)maxblocks * 4);
    for (auto &t : tests) {
        printf("%-22s |", t.name);
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int blocks = cus * wps;                      // 256-thread block = 4 waves = 1 wave per SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, ticks, 1u);
            hipDeviceSynchronize();
            const int reps = 20;
            hipEventRecord(e0, 0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, ticks, 1u);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), ticks, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < blocks * 4; ++i) avg += (double)h[i]; avg /= blocks * 4;
            const double instr = (double)ITERS * 8 * t.per_body;
            const double laneops = (double)reps * blocks * 256.0 * instr;
            // a wave's elapsed ticks / its instruction count, times waves per SIMD sharing the pipe
            printf(" %6.2f cyc  %6.2f T/s %5.0f MHz |", avg / instr / wps * 1.0, laneops / (ms * 1e-3) / 1e12, avg / (ms * 1e3 / reps));
        }
        printf("\n");
    }
    return 0;
}
