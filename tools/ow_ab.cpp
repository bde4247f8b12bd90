// tools/ow_ab.cpp -- A/B of the one-workgroup transforms against the two-pass kernels through the C ABI: bitwise equality
// of the zero-padded forward transform at 16K / 32K / 64K points (odd batches, random u32 input), then timing of both
// forms.  Build: hipcc -O2 tools/ow_ab.cpp -Iinclude -Lcuhe_amd/lib -lcuhe_hip -Wl,-rpath,$PWD/cuhe_amd/lib -o cuhe_amd/lib/ow_ab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cuhe_hip.h"

#define CK(x) do { int r_ = (x); if (r_) { printf("FAIL %s -> %d: %s\n", #x, r_, cuhe_hip_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint64_t sm(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char **argv) {
    const int big = argc > 1 ? atoi(argv[1]) : 4096;          // transforms per timed call at 64K points
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    // equality against the two-pass kernels: every length at a small odd batch (mode 3: one-workgroup form whatever the row
    // count), and 64K / 32K points at batches that take the persistent form (mode 1; 301 / 1101 rows: padding items in the last group of 8)
    struct Case { int len, batch, mode, r64; };
    for (Case cs : {Case{16384, 37, 2, 0}, Case{32768, 37, 2, 0}, Case{65536, 37, 2, 1}, Case{65536, 301, 1, 2}, Case{65536, 301, 1, 1}, Case{32768, 600, 1, 1}, Case{32768, 1101, 1, 3}, Case{32768, 1101, 1, 1}, Case{16384, 1100, 1, 0}}) {
        const int len = cs.len, batch = cs.batch;
        CK(cuhe_hip_ntt_prepare(len, 0));
        std::vector<uint32_t> h((size_t)batch * len / 2);
        uint64_t s = len + batch;
        for (auto &v : h) v = (uint32_t)sm(s);
        h[0] = 0xffffffffu; h[1] = 0; h[2] = 1;
        uint32_t *dx; uint64_t *dA, *dB;
        HK(hipMalloc(&dx, h.size() * 4)); HK(hipMalloc(&dA, (size_t)batch * len * 8)); HK(hipMalloc(&dB, (size_t)batch * len * 8));
        HK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        HK(hipMemset(dA, 0xAA, (size_t)batch * len * 8)); HK(hipMemset(dB, 0x55, (size_t)batch * len * 8));
        CK(cuhe_hip_set_onewg(0, 0));
        CK(cuhe_hip_ntt_fwd_batched(dA, dx, len, batch, len / 2, 0, nullptr));
        CK(cuhe_hip_set_onewg(cs.mode, cs.r64));
        for (int rep = 0; rep < 2; ++rep) CK(cuhe_hip_ntt_fwd_batched(dB, dx, len, batch, len / 2, 0, nullptr));
        HK(hipDeviceSynchronize());
        std::vector<uint64_t> a((size_t)batch * len), b((size_t)batch * len);
        HK(hipMemcpy(a.data(), dA, a.size() * 8, hipMemcpyDeviceToHost)); HK(hipMemcpy(b.data(), dB, b.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
        printf("len %d batch %d mode %d/%d: %zu mismatches%s\n", len, batch, cs.mode, cs.r64, bad, bad ? "" : "  (bit-identical)");
        if (bad) printf("   first at row %zu index %zu: two-pass %016llx one-wg %016llx\n", first / len, first % len, (unsigned long long)a[first], (unsigned long long)b[first]);
        hipFree(dx); hipFree(dA); hipFree(dB);
    }
    for (int len : {65536, 32768, 16384}) {
        const int batch = (int)((long)big * 65536 / len);
        uint32_t *dx; uint64_t *dA;
        HK(hipMalloc(&dx, (size_t)batch * len / 2 * 4)); HK(hipMalloc(&dA, (size_t)batch * len * 8));
        std::vector<uint32_t> h((size_t)batch * len / 2);
        uint64_t s = 77 + len;
        for (auto &v : h) v = (uint32_t)sm(s);
        HK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (int mode : {0, 2, 1, 0, 2, 1}) {
            if (len == 16384 && mode >= 2) continue;              // (the forms differ at 32K and 64K points only)
            CK(cuhe_hip_set_onewg(mode ? 1 : 0, mode == 2 ? 1 : mode == 1 ? (len == 32768 ? 3 : 2) : 0));
            float p1 = 0, p2 = 0, tot = 0;
            CK(cuhe_hip_time_ntt_fwd(dA, dx, len, batch, 2, 0, nullptr, &p1, &p2, &tot));       // warm
            CK(cuhe_hip_time_ntt_fwd(dA, dx, len, batch, iters, 0, nullptr, &p1, &p2, &tot));
            const double per = tot / iters * 1e-3 / batch;
            printf("len %d batch %d %s: %.4f ms per call, %.3f M transforms/s, HBM-roofline frac %.4f  (passes %.4f + %.4f ms)\n", len, batch,
                   mode == 0 ? "two-pass" : mode == 2 ? "one-wg  " : len >= 32768 ? "one-wg persistent" : "one-wg  ", tot / iters, 1e-6 / per, 10.0 * len / per / 8e12, p1 / iters, p2 / iters);
        }
        hipFree(dx); hipFree(dA);
    }
    return 0;
}
