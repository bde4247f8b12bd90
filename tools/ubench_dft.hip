// ubench_dft.hip -- issue efficiency of the in-register DFT bodies (no memory traffic):
// runs dft_regs<N> ITERS times on register-resident data at several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../cuhe_amd/csrc/ntt_kernels.cuh"
using namespace cuhe;
#define ITERS 64

template <int N>
__global__ __launch_bounds__(256) void kern(u64 *out, u64 seed) {
    u64 x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = canon(seed * (2 * i + 3) + threadIdx.x * 977 + blockIdx.x);
    for (int it = 0; it < ITERS; ++it) dft_regs<N, false>(x);
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int N>
void run(u64 *out, int cus, int wgs_per_cu, double valu_per_dft) {
    int blocks = cus * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<N>, dim3(blocks), dim3(256), 0, 0, out, 12345ULL);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern<N>, dim3(blocks), dim3(256), 0, 0, out, 12345ULL);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double dfts = 3.0 * blocks * 256.0 * ITERS;
    double pts = dfts * N;
    printf("dft<%d> %d WG/CU (%d waves/SIMD): %.3f G pts/s  %.1f ns/DFT/thread  => %.1f VALU lanes/clk/CU @2.4GHz (%.0f VALU/DFT)\n", N, wgs_per_cu, wgs_per_cu,
           pts / (ms * 1e-3) / 1e9, ms * 1e6 / (3.0 * ITERS), dfts * valu_per_dft / (ms * 1e-3) / cus / 2.4e9, valu_per_dft);
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    u64 *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    double v64 = argc > 1 ? atof(argv[1]) : 3852, v32 = argc > 2 ? atof(argv[2]) : 1548, v16 = argc > 3 ? atof(argv[3]) : 600;
    for (int occ : {1, 2, 3, 4}) run<64>(out, cus, occ, v64);
    for (int occ : {1, 2, 4, 8}) run<32>(out, cus, occ, v32);
    for (int occ : {1, 2, 4, 8}) run<16>(out, cus, occ, v16);
    return 0;
}
