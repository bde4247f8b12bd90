// ubench_clock.hip -- effective shader clock under the NTT instruction mix: per-wave cycle counts (s_memtime)
// of the in-register DFT64 body vs wall time, at 2 waves/SIMD, with and without concurrent HBM streaming.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../cuhe_amd/csrc/ntt_kernels.cuh"
using namespace cuhe;

__global__ __launch_bounds__(256, 2) void body(u64 *out, unsigned long long *cyc, int iters) {
    u64 x[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) x[i] = canon(0x9E3779B97F4A7C15ULL * (2 * i + 3) + threadIdx.x * 977 + blockIdx.x);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) dft_regs<64, false>(x);
    long long t1 = clock64();
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}
__global__ void wallclk(unsigned long long *o) { o[0] = wall_clock64(); o[1] = clock64(); }

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
    printf("CUs %d, clockRate %d kHz, wallClockRate %d kHz\n", cus, prop.clockRate, wcr);
    u64 *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8); hipMalloc(&cyc, 8);
    for (int iters : {1, 4, 16, 64}) {
        int blocks = cus * 2;
        hipMemset(cyc, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(body, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(body, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double waves = blocks * 4.0;
        double cyc_per_wave = h / waves;
        printf("iters %2d: wall %.1f us, s_memtime ticks per wave %.0f  => ticks/us %.1f ; VALU 4199/iter: %.2f ticks per VALU instr per wave\n", iters, ms * 1e3,
               cyc_per_wave, cyc_per_wave / (ms * 1e3), cyc_per_wave / (4199.0 * iters));
    }
    return 0;
}
