// ubench_mfma_dft.hip -- the premise of a hybrid transform stage on the matrix cores: do int8 MFMA
// instructions of SOME waves issue beside the field arithmetic (VALU) of OTHER waves of the same SIMD without slowing it,
// under the chip's power limit?  Two workgroups of 512 threads per CU (the occupancy of the 16K-point one-workgroup
// transforms); in every workgroup the waves 0-3 run a dependent-free stream of 16-point shift-only transforms (dft_regs<16>,
// the instruction mix of the register stages), the waves 4-7 a stream of v_mfma_i32_16x16x64_i8 on four accumulator
// tiles -- so every SIMD holds VALU waves and MFMA waves.  Modes: VALU waves alone, MFMA waves alone, both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_dft.hip -o tools/ubench_mfma_dft
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../cuhe_amd/csrc/ntt_kernels.cuh"
using namespace cuhe;
typedef int v4i __attribute__((ext_vector_type(4)));
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// mode bit 0: the VALU waves work; bit 1: the MFMA waves work
__global__ __launch_bounds__(512, 4)
void k_mix(u64 *out, const u64 *in, int iters_valu, int iters_mfma, int mode) {
    const int t = threadIdx.x, wave = t >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        u64 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = in[(blockIdx.x * 512 + t) * 16 + i];
        for (int it = 0; it < iters_valu; ++it) {
            dft_regs<16, false>(x);
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]));
        }
        u64 s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s ^= x[i];
        out[blockIdx.x * 512 + t] = s;
    } else {
        if (!(mode & 2)) return;
        v4i a, b, acc[4];
        const int *p = (const int *)in + (t & 255) * 8;
        a = v4i{p[0], p[1], p[2], p[3]}; b = v4i{p[4], p[5], p[6], p[7]};
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
        }
        v4i s = acc[0] + acc[1] + acc[2] + acc[3];
        out[blockIdx.x * 512 + t] = (u64)(unsigned)(s[0] ^ s[1] ^ s[2] ^ s[3]);
    }
}

int main(int argc, char **argv) {
    const int iv = argc > 1 ? atoi(argv[1]) : 4000, im = argc > 2 ? atoi(argv[2]) : 12000;
    int dev = 0; hipDeviceProp_t pr; HK(hipGetDeviceProperties(&pr, dev));
    const int grid = pr.multiProcessorCount * 2;
    u64 *in, *out;
    HK(hipMalloc(&in, (size_t)grid * 512 * 16 * 8)); HK(hipMalloc(&out, (size_t)grid * 512 * 8));
    HK(hipMemset(in, 0x5a, (size_t)grid * 512 * 16 * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // a 16-point transform: 32 butterflies (add + sub) and 17 non-trivial shifts
    const double points_per_valu_iter = 16.0 * grid * 256, macs_per_mfma_iter = 4.0 * 16 * 16 * 64 * grid * 4;     // 256 VALU lanes, 4 MFMA waves per workgroup
    auto run = [&](int mode, int a, int b) -> double {
        hipLaunchKernelGGL(k_mix, dim3(grid), dim3(512), 0, 0, out, in, a, b, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mix, dim3(grid), dim3(512), 0, 0, out, in, a, b, mode);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    // calibrate both streams to ~3 ms stand-alone, then run them together: 3 ms = the pipes are independent, 6 ms = they exclude each other
    int a = iv, b = im;
    a = (int)(a * 3.0 / run(1, a, b)); b = (int)(b * 3.0 / run(2, a, b));
    for (int rep = 0; rep < 3; ++rep) {
        const double t1 = run(1, a, b), t2 = run(2, a, b), t3 = run(3, a, b);
        printf("VALU waves alone %.3f ms (%.1f G points/s of 16-point transforms) | MFMA waves alone %.3f ms (%.2f POPS int8) | together %.3f ms"
               "  => overlap %.0f %% of the shorter stream hidden; rates together: %.1f G points/s, %.2f POPS\n",
               t1, a * points_per_valu_iter / (t1 * 1e-3) / 1e9, t2, 2 * b * macs_per_mfma_iter / (t2 * 1e-3) / 1e15, t3,
               100.0 * (t1 + t2 - t3) / (t1 < t2 ? t1 : t2), a * points_per_valu_iter / (t3 * 1e-3) / 1e9, 2 * b * macs_per_mfma_iter / (t3 * 1e-3) / 1e15);
    }
    // a quarter of the MFMA work beside the full VALU stream: what a hybrid stage would ask for
    for (int frac : {2, 4}) {
        const double t1 = run(1, a, b / frac), t3 = run(3, a, b / frac);
        printf("full VALU stream + 1/%d of the MFMA stream: %.3f ms against %.3f ms for the VALU stream alone (%+.1f %%)\n", frac, t3, t1, 100.0 * (t3 - t1) / t1);
    }
    return 0;
}
