// sample_profiler.hpp -- a minimal sampling profiler for the host side of the test clients (no perf / gdb in the image):
// a sampler thread signals every thread of the process (/proc/self/task) every ~200 us of WALL time, the handler keeps the
// program counters of the interrupted thread's stack (a thread blocked in a futex shows up as such: busy against idle);
// report() prints, per shared object, the hottest return addresses as object-relative offsets, ready for
//   addr2line -f -C -e <object> <offset> ...        (tools/resolve_samples.py does that on the build machine).
// Diagnostics only: used by test_prince_flow --profile to see where the scheduler's workers spend their host time.
#pragma once
#include <dlfcn.h>
#include <execinfo.h>
#include <dirent.h>
#include <signal.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
namespace sample_profiler {
constexpr int kDepth = 28, kMax = 200000;
static void *g_pc[kMax][kDepth];
static int g_tid[kMax];
static volatile int g_n = 0;
static void handler(int) {
	const int i = __sync_fetch_and_add(&g_n, 1);
	if (i >= kMax) return;
	g_tid[i] = (int)syscall(SYS_gettid);
	void *buf[kDepth + 2];
	const int n = backtrace(buf, kDepth + 2);
	for (int k = 0; k < kDepth; ++k) g_pc[i][k] = k + 2 < n ? buf[k + 2] : nullptr;      // skip the handler and the signal trampoline
}
static std::atomic<bool> g_run{false};
static std::thread g_sampler;
inline void start() {
	void *warm[4]; backtrace(warm, 4);          // loads libgcc's unwinder outside the handler
	g_n = 0;                                    // (a new profile per start: test_prince_flow --repeat N --profile reports every block on its own)
	struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = handler; sa.sa_flags = SA_RESTART;
	sigaction(SIGUSR2, &sa, nullptr);
	g_run = true;
	g_sampler = std::thread([] {
		const pid_t pid = getpid(), self = (pid_t)syscall(SYS_gettid);
		std::vector<pid_t> tids; int round = 0;
		while (g_run) {
			if (round++ % 64 == 0) {            // the thread list, now and then
				tids.clear();
				if (DIR *d = opendir("/proc/self/task")) {
					while (dirent *e = readdir(d)) { const pid_t t = (pid_t)atoi(e->d_name); if (t > 0 && t != self) tids.push_back(t); }
					closedir(d);
				}
			}
			for (pid_t t : tids) syscall(SYS_tgkill, pid, t, SIGUSR2);
			struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr);
		}
	});
}
inline void stop() { g_run = false; if (g_sampler.joinable()) g_sampler.join(); }
// only the samples whose stack passes through `marker_obj` (e.g. "libcuHE.so": the scheduler's workers and the client thread);
// leaf = the innermost frame, by symbol where the object exports one; then, per leaf symbol, the innermost frame INSIDE our own objects
inline void report(FILE *f, const char *marker_obj = "libcuHE.so", int top = 30) {
	const int n = std::min((int)g_n, kMax);
	std::map<std::string, int> leaf, ours, pair;
	int kept = 0;
	for (int i = 0; i < n; ++i) {
		bool mine = false; std::string l, o;
		for (int k = 0; k < kDepth && g_pc[i][k]; ++k) {
			Dl_info di;
			if (!dladdr(g_pc[i][k], &di) || !di.dli_fname) continue;
			const char *b = strrchr(di.dli_fname, '/');
			const std::string obj = b ? b + 1 : di.dli_fname;
			char buf[256];
			snprintf(buf, sizeof buf, "%s:%s", obj.c_str(), di.dli_sname ? di.dli_sname : "?");
			if (k == 0) l = buf;
			if (o.empty() && (obj == "libcuHE.so" || obj == "libcuhe_hip.so")) {
				snprintf(buf, sizeof buf, "%s+0x%lx", obj.c_str(), (unsigned long)((char *)g_pc[i][k] - (char *)di.dli_fbase));
				o = buf;
			}
			if (obj == marker_obj) mine = true;
		}
		if (!mine) continue;
		++kept; ++leaf[l]; ++ours[o]; ++pair[l + "  <-  " + o];
	}
	fprintf(f, "samples %d (every thread, every ~200 us of wall time), %d with a frame in %s\n", n, kept, marker_obj);
	auto dump = [&](const char *title, std::map<std::string, int> &m) {
		std::vector<std::pair<int, std::string>> v;
		for (auto &e : m) v.push_back({e.second, e.first});
		std::sort(v.begin(), v.end(), [](auto &x, auto &y) { return x.first > y.first; });
		fprintf(f, "%s\n", title);
		for (int i = 0; i < (int)v.size() && i < top; ++i) fprintf(f, "  %6d  %5.1f %%  %s\n", v[i].first, 100.0 * v[i].first / std::max(kept, 1), v[i].second.c_str());
	};
	// per thread: the innermost frame inside libcuHE.so (the call site in the C++ layer the sample sits under), so that the ONE worker that
	// issues the batches can be read on its own (tools/resolve_samples.py adds file:line)
	std::map<int, std::map<std::string, int>> perThread; std::map<int, int> perThreadTotal;
	for (int i = 0; i < n; ++i) {
		std::string site;
		for (int k = 0; k < kDepth && g_pc[i][k] && site.empty(); ++k) {
			Dl_info di;
			if (!dladdr(g_pc[i][k], &di) || !di.dli_fname) continue;
			const char *b = strrchr(di.dli_fname, '/');
			if (std::string(b ? b + 1 : di.dli_fname) != marker_obj) continue;
			char buf[256];
			snprintf(buf, sizeof buf, "%s+0x%lx", marker_obj, (unsigned long)((char *)g_pc[i][k] - (char *)di.dli_fbase));
			site = buf;
		}
		if (site.empty()) continue;
		++perThread[g_tid[i]][site]; ++perThreadTotal[g_tid[i]];
	}
	for (auto &t : perThread) {
		std::vector<std::pair<int, std::string>> v;
		for (auto &e : t.second) v.push_back({e.second, e.first});
		std::sort(v.begin(), v.end(), [](auto &x, auto &y) { return x.first > y.first; });
		fprintf(f, "thread %d: %d samples; innermost call site in %s:\n", t.first, perThreadTotal[t.first], marker_obj);
		for (int i = 0; i < (int)v.size() && i < 22; ++i) fprintf(f, "  %6d  %5.1f %%  %s\n", v[i].first, 100.0 * v[i].first / std::max(perThreadTotal[t.first], 1), v[i].second.c_str());
	}
	dump("LEAF symbol:", leaf);
	dump("innermost frame in libcuHE.so / libcuhe_hip.so (offsets: tools/resolve_samples.py):", ours);
	dump("leaf <- innermost own frame:", pair);
}
}  // namespace sample_profiler
