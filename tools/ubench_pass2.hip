// ubench_pass2.hip -- is ntt_pass2 bandwidth-bound?  Same grid / access pattern with (a) no arithmetic,
// (b) the real kernel, (c) a plain 16 B/lane streaming copy of the same byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../cuhe_amd/csrc/ntt_kernels.cuh"
using namespace cuhe;

template <int LG>
__global__ __launch_bounds__(256, 2) void pass2_pattern(u64 *dst, const u64 *scratch, int nbatch) {
    constexpr int L = 1 << LG, N1 = L / 64;
    int batch, tile;
    xcd_map(N1 / 256, batch, tile);
    if (batch >= nbatch) return;
    const int k1 = tile * 256 + threadIdx.x;
    const u64 *in = scratch + (long)batch * L + k1;
    u64 x[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) x[j] = in[j * N1];
    u64 *d = dst + (long)batch * L + k1;
#pragma unroll
    for (int k2 = 0; k2 < 64; ++k2) d[k2 * N1] = x[bitrev<64>(k2)] + 1;
}
// the real pass-2 body with an optional start-up skew for every other workgroup (lockstep diagnosis)
template <int LG>
__global__ __launch_bounds__(256, 2) void pass2_skew(u64 *dst, const u64 *scratch, const u64 *T2, int nbatch, int skew_cycles) {
    constexpr int L = 1 << LG, N1 = L / 64;
    if (skew_cycles && ((blockIdx.x >> 8) & 1)) {
        long long t0 = clock64();
        while (clock64() - t0 < skew_cycles) __builtin_amdgcn_s_sleep(8);
    }
    int batch, tile;
    xcd_map(N1 / 256, batch, tile);
    if (batch >= nbatch) return;
    const int k1 = tile * 256 + threadIdx.x;
    const u64 *in = scratch + (long)batch * L + k1;
    const u64 *tw = T2 + k1;
    u64 x[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) { u64 v = in[j * N1]; if (j != 0) v = mulp(v, tw[j * N1]); x[j] = v; }
    dft_regs<64, false>(x);
    u64 *d = dst + (long)batch * L + k1;
#pragma unroll
    for (int k2 = 0; k2 < 64; ++k2) d[k2 * N1] = x[bitrev<64>(k2)];
}
// ablations of the real body: LOAD / MUL / STORE phases switched off individually
template <int LG, bool LOAD, bool MUL, bool STORE>
__global__ __launch_bounds__(256, 2) void pass2_abl(u64 *dst, const u64 *scratch, const u64 *T2, int nbatch) {
    constexpr int L = 1 << LG, N1 = L / 64;
    int batch, tile;
    xcd_map(N1 / 256, batch, tile);
    if (batch >= nbatch) return;
    const int k1 = tile * 256 + threadIdx.x;
    const u64 *in = scratch + (long)batch * L + k1;
    const u64 *tw = T2 + k1;
    u64 x[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        u64 v = LOAD ? in[j * N1] : canon((u64)k1 * (2 * j + 1) * 0x9E3779B97F4A7C15ULL);
        if (MUL && j != 0) v = mulp(v, LOAD ? tw[j * N1] : canon(v ^ 0x1234567));
        x[j] = v;
    }
    dft_regs<64, false>(x);
    u64 *d = dst + (long)batch * L + k1;
    if (STORE) {
#pragma unroll
        for (int k2 = 0; k2 < 64; ++k2) d[k2 * N1] = x[bitrev<64>(k2)];
    } else {
        u64 acc = 0;
#pragma unroll
        for (int k2 = 0; k2 < 64; ++k2) acc ^= x[k2];
        if (acc == 0x123456789ULL) d[0] = acc;
    }
}
__global__ __launch_bounds__(256) void stream_copy(ulonglong2 *dst, const ulonglong2 *src, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    const int LG = 16, L = 1 << LG, NB = 256;
    u64 *a, *b, *t2;
    size_t bytes = (size_t)NB * L * 8;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&t2, (size_t)L * 8);
    hipMemset(a, 1, bytes); hipMemset(t2, 1, (size_t)L * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %.3f ms per %d transforms  (%.2f TB/s for read+write of %zu MiB)\n", name, ms, NB, 2.0 * bytes / (ms * 1e-3) / 1e12, 2 * bytes >> 20);
    };
    const int grid = (NB / 8) * 8 * (L / 64 / 256);
    timeit("pass2 access pattern only", [&] { hipLaunchKernelGGL(pass2_pattern<16>, dim3(grid), dim3(256), 0, 0, b, a, NB); });
    timeit("ntt_pass2<16,false> (real)", [&] { hipLaunchKernelGGL((ntt_pass2<16, false>), dim3(grid), dim3(256), 0, 0, (void *)b, a, t2, (long)L, NB, L, nullptr, nullptr, 0); });
    for (int skew : {0})
        timeit(skew ? "real body, odd WGs skewed" : "real body, no skew", [&] { hipLaunchKernelGGL(pass2_skew<16>, dim3(grid), dim3(256), 0, 0, b, a, t2, NB, skew); });
    timeit("abl: load+mul+dft+store", [&] { hipLaunchKernelGGL((pass2_abl<16, true, true, true>), dim3(grid), dim3(256), 0, 0, b, a, t2, NB); });
    timeit("abl: load+dft+store (no mul)", [&] { hipLaunchKernelGGL((pass2_abl<16, true, false, true>), dim3(grid), dim3(256), 0, 0, b, a, t2, NB); });
    timeit("abl: load+mul+dft (no store)", [&] { hipLaunchKernelGGL((pass2_abl<16, true, true, false>), dim3(grid), dim3(256), 0, 0, b, a, t2, NB); });
    timeit("abl: mul+dft+store (no load)", [&] { hipLaunchKernelGGL((pass2_abl<16, false, true, true>), dim3(grid), dim3(256), 0, 0, b, a, t2, NB); });
    timeit("abl: mul+dft (no load/store)", [&] { hipLaunchKernelGGL((pass2_abl<16, false, true, false>), dim3(grid), dim3(256), 0, 0, b, a, t2, NB); });
    timeit("abl: dft only", [&] { hipLaunchKernelGGL((pass2_abl<16, false, false, false>), dim3(grid), dim3(256), 0, 0, b, a, t2, NB); });
    timeit("stream copy 16B/lane", [&] { hipLaunchKernelGGL(stream_copy, dim3(256 * 8), dim3(256), 0, 0, (ulonglong2 *)b, (const ulonglong2 *)a, (long)(bytes / 16)); });
    return 0;
}
