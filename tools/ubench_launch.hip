// ubench_launch.hip -- what a kernel launch costs the HOST on this box when T threads launch at the same time, each on
// its own stream: the floor under the gate scheduler (cuhe_amd/cxx/Scheduler.cpp), whose ~68k launches per PRINCE block
// are issued by a few worker threads.   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_launch tools/ubench_launch.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
    const int N = 20000;
    for (int T : {1, 2, 3, 4, 6, 8, 12}) {
        std::vector<hipStream_t> st(T);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int mode = 0; mode < 2; ++mode) {          // 0: launches only, 1: an event record after every third launch
            std::vector<std::thread> th;
            std::vector<hipEvent_t> ev(T);
            for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
            const auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (int i = 0; i < N; ++i) {
                        hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st[t], (int *)nullptr);
                        if (mode && i % 3 == 2) hipEventRecord(ev[t], st[t]);
                    }
                });
            for (auto &x : th) x.join();
            const double host = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            hipDeviceSynchronize();
            const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("T=%2d %s: %.2f us of host time per launch and thread, %.0f k launches/s enqueued, %.0f k/s executed\n", T,
                   mode ? "launch + event/3" : "launch only     ", host / N * 1e6, T * N / host / 1e3, T * N / all / 1e3);
            for (auto &e : ev) hipEventDestroy(e);
        }
        for (auto &s : st) hipStreamDestroy(s);
    }
    return 0;
}
