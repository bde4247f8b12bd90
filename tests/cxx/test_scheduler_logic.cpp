// test_scheduler_logic.cpp -- the gate scheduler (cuhe_amd/cxx/Scheduler.cpp) WITHOUT a GPU: the translation unit is compiled
// against a mock of the few C-ABI calls it makes (streams, events, blocks), and random gate programs are recorded on it.
//
// What is checked, for 1 ... 8 devices, every batch policy, with and without the batch runner:
//  * host order: a task is issued only after every task it depends on (last writer of what it reads; last writer and all
//    readers since of what it writes) has been issued;
//  * DEVICE order: the mock keeps a vector clock per stream -- a launch stamps the task with its stream's clock, an event record
//    copies the clock, a stream wait joins it -- so "the producer's work precedes the consumer's work on the GPU" is checked
//    through exactly the lazily recorded per-stream events the scheduler relies on, across streams and across devices;
//  * batches: members share kind, key and device, never exceed the cap, and every gate runs exactly once;
//  * BLOCKS: gates take device blocks inside tasks (taskAlloc), every gate that touches the polynomial uses its block on its stream,
//    release-only tasks (sched::kReleaseOnly: they wait for nothing and enqueue nothing) hand the blocks back (taskFree) -- whoever
//    gets such a block next must be ordered, on the device, behind EVERY use it had, whichever streams those ran on;
//  * a task runs on a stream of ITS device, workers are per device, drain / wait / waitNode / stop / restart work, nothing leaks
//    (every Task and Node is deleted: counted through the references the test holds).
// Built and run by tests/test_scheduler_logic.py (plain g++, also under -fsanitize=thread when the toolchain has it).
#include <atomic>
#include <cassert>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <string>
#include <vector>

// ---------------------------------------------------------------- mock of the C ABI (only what Scheduler.cpp uses)
#include "../../include/cuhe_hip.h"
namespace mock {
typedef std::map<int, long> Clock;                    // stream id -> number of launches of that stream that precede
struct Stream { int id, dev; Clock vc; long launches = 0; };
struct Event { Clock vc; bool recorded = false; };
std::mutex m;
std::vector<Stream *> streams;
int numGpus = 1;
std::atomic<long> eventRecords{0}, streamWaits{0}, mallocs{0}, frees{0};
unsigned long long generation = 1;
void *recycled = nullptr; bool recycle = false;
void join(Clock &a, const Clock &b) { for (auto &kv : b) if (a[kv.first] < kv.second) a[kv.first] = kv.second; }
// a task's device work enqueued on `stream`: returns the stamp (the stream's clock including this launch)
Clock launch(void *stream) {
	std::lock_guard<std::mutex> lk(m);
	Stream *s = (Stream *)stream;
	s->vc[s->id] = ++s->launches;
	return s->vc;
}
int deviceOf(void *stream) { return ((Stream *)stream)->dev; }
int idOf(void *stream) { return ((Stream *)stream)->id; }
}  // namespace mock
extern "C" {
const char *cuhe_hip_last_error(void) { return "mock"; }
int cuhe_hip_is_initialised(void) { return 1; }
int cuhe_hip_num_gpus(void) { return mock::numGpus; }
unsigned long long cuhe_hip_generation(void) { return mock::generation; }
int cuhe_hip_stream_create(int dev, void **out) {
	std::lock_guard<std::mutex> lk(mock::m);
	mock::Stream *s = new mock::Stream; s->id = (int)mock::streams.size(); s->dev = dev; mock::streams.push_back(s); *out = s; return 0;
}
int cuhe_hip_event_create(int, void **out) { *out = new mock::Event; return 0; }
int cuhe_hip_event_record(int dev, void *ev, void *st) {
	std::lock_guard<std::mutex> lk(mock::m);
	mock::Stream *s = (mock::Stream *)st;
	if (s->dev != dev) { printf("event recorded with device %d on a stream of device %d\n", dev, s->dev); exit(3); }
	((mock::Event *)ev)->vc = s->vc; ((mock::Event *)ev)->recorded = true; ++mock::eventRecords; return 0;
}
int cuhe_hip_stream_wait_event(int dev, void *st, void *ev) {
	std::lock_guard<std::mutex> lk(mock::m);
	mock::Stream *s = (mock::Stream *)st;
	if (s->dev != dev) { printf("stream wait with device %d on a stream of device %d\n", dev, s->dev); exit(3); }
	if (!((mock::Event *)ev)->recorded) { printf("wait for an event that was never recorded\n"); exit(3); }
	mock::join(s->vc, ((mock::Event *)ev)->vc); ++mock::streamWaits; return 0;
}
int cuhe_hip_event_sync(int, void *) { return 0; }
int cuhe_hip_device_sync(int) { return 0; }
int cuhe_hip_pin_thread_to_device(int) { return 0; }
// (mock::recycle: the next cuhe_hip_free keeps the pointer and the next cuhe_hip_malloc hands the SAME address out again whatever the size --
// what a real allocator may do at any time -- for the stale-size check of crossDeviceBlocks)
void *cuhe_hip_malloc(int, size_t bytes) {
	++mock::mallocs;
	if (mock::recycled) { void *p = mock::recycled; mock::recycled = nullptr; return p; }
	return malloc(bytes < 65536 ? 65536 : bytes);
}
int cuhe_hip_free(int, void *p) { ++mock::frees; if (mock::recycle && !mock::recycled) { mock::recycled = p; mock::recycle = false; return 0; } free(p); return 0; }
int cuhe_hip_alloc_counters(long long *out4) { for (int i = 0; i < 4; ++i) out4[i] = 0; return 0; }
}

// ---------------------------------------------------------------- the unit under test
#include "../../cuhe_amd/cxx/Scheduler.cpp"
// (CuHE.h declares the polynomial classes; the test records on nodes without objects, so none of their members is needed)

using namespace cuHE;

namespace {
struct Gate {
	int id, dev, kind; long key;
	int w = -1; std::vector<int> rd; bool allocs = false;      // block model: the node written, the nodes read, "takes a block for w if it has none"
	std::vector<int> mustFollow;                       // gate ids, from the test's own model of the dependency rule
	std::atomic<int> runs{0};
	mock::Clock stamp; int stream = -1;
	bool blocking = false;                             // sched::kHostBlocking or keep = true: must not run on a worker that takes groups when the device has others
};
// streams that ran the batch runner / that ran host-blocking gates, per program (round 6: Scheduler.h, kHostBlocking)
std::set<int> groupStreams, blockingStreams;
std::vector<Gate *> gates;
std::mutex gm;
std::atomic<long> failures{0}, batchesSeen{0}, batchedGates{0};
std::atomic<int> maxBatchSeen{0};
void fail(const char *what, int a = 0, int b = 0) { ++failures; fprintf(stderr, "FAIL: %s (%d, %d)\n", what, a, b); }
// block model (program(..., blocks = true)): the block a node holds (touched by the node's gates only, which the scheduler orders on the host),
// and every use a block has had since it was handed out: (stream id, launch number on that stream)
bool blocksOn = false;
std::vector<void *> blk;
std::map<void *, std::vector<std::pair<int, long>>> uses;
std::atomic<long> blocksTaken{0}, blocksReused{0}, blocksReleased{0};

void runGate(Gate *g, void *stream) {
	if (mock::deviceOf(stream) != g->dev) fail("gate ran on a stream of another device", g->id, mock::deviceOf(stream));
	if (!sched::inWorker() || sched::workerStream() != stream) fail("workerStream() is not the task's stream", g->id);
	// host order + device order of everything it must follow
	for (int d : g->mustFollow) {
		Gate *p = gates[d];
		if (p->runs.load() != 1) { fail("dependency not issued before its consumer", g->id, d); continue; }
	}
	void *took = nullptr;
	if (blocksOn && g->allocs && !blk[g->w]) { took = sched::taskAlloc(g->dev, 4096); if (!took) fail("taskAlloc returned nothing", g->id); }      // (its waits precede the launch)
	const mock::Clock now = mock::launch(stream);
	if (blocksOn) {
		std::lock_guard<std::mutex> lk(gm);
		const int sid = mock::idOf(stream);
		if (took) {
			++blocksTaken;
			auto &u = uses[took];
			if (!u.empty()) ++blocksReused;
			for (auto &e : u) { auto it = now.find(e.first); if ((it == now.end() ? 0 : it->second) < e.second) fail("a block was handed out before one of its uses is ordered behind", g->id, e.first); }
			u.clear();
			blk[g->w] = took;
		}
		if (blk[g->w]) uses[blk[g->w]].push_back({sid, now.at(sid)});
		for (int r : g->rd) if (r != g->w && blk[r]) uses[blk[r]].push_back({sid, now.at(sid)});
	}
	{
		std::lock_guard<std::mutex> lk(gm);
		for (int d : g->mustFollow) {
			Gate *p = gates[d];
			if (p->stream < 0) continue;
			auto it = now.find(p->stream);
			const long seen = it == now.end() ? 0 : it->second;
			const long need = p->stamp.at(p->stream);
			if (seen < need) fail("device order: the producer's launch does not precede the consumer's", g->id, d);
		}
		g->stamp = now; g->stream = mock::idOf(stream);
		if (g->blocking) blockingStreams.insert(g->stream);
	}
	if (g->runs.fetch_add(1) != 0) fail("gate ran twice", g->id);
}
// a release-only gate: no launch, the node's block goes back
void runRelease(Gate *g, void *stream) {
	if (mock::deviceOf(stream) != g->dev) fail("release ran on a stream of another device", g->id);
	for (int d : g->mustFollow) if (gates[d]->runs.load() != 1) fail("dependency not issued before the release", g->id, d);
	void *p;
	{ std::lock_guard<std::mutex> lk(gm); p = blk[g->w]; blk[g->w] = nullptr; }
	if (p) { ++blocksReleased; if (!sched::taskFree(g->dev, p)) fail("taskFree refused a block taskAlloc handed out", g->id); }
	if (g->runs.fetch_add(1) != 0) fail("gate ran twice", g->id);
}
// node -> gate of the recorded batchable task, to find the members inside the batch runner
std::map<sched::Node *, std::vector<Gate *>> pendingBySubject;
void batchRunner(int kind, sched::Node *const *subjects, sched::Node *const *, sched::Node *const *, int count, void *stream) {
	++batchesSeen; batchedGates += count;
	int seen = maxBatchSeen.load(); while (count > seen && !maxBatchSeen.compare_exchange_weak(seen, count)) {}
	if (count < 2) fail("batch runner called for fewer than two gates", count);
	{ std::lock_guard<std::mutex> lk(gm); groupStreams.insert(mock::idOf(stream)); }
	Gate *first = nullptr;
	for (int i = 0; i < count; ++i) {
		Gate *g;
		{
			std::lock_guard<std::mutex> lk(gm);
			auto &v = pendingBySubject[subjects[i]];
			if (v.empty()) { fail("batch member without a recorded gate"); continue; }
			g = v.front(); v.erase(v.begin());            // gates on one subject are ordered: the oldest is the one that is ready
		}
		if (g->kind != kind) fail("batch member of another kind", g->id);
		if (!first) first = g;
		else if (g->key != first->key || g->dev != first->dev) fail("batch mixes keys or devices", g->id, first->id);
		runGate(g, stream);
	}
}

// one random program: `nodes` polynomials spread over `ndev` devices, `ngates` gates
void program(int ndev, int nodesN, int ngates, unsigned seed, bool batches, int policyNo, int cap, bool blocks = false, int workers = 3) {
	mock::numGpus = ndev;
	groupStreams.clear(); blockingStreams.clear();
	blocksOn = blocks; blk.assign(nodesN, nullptr); uses.clear();
	setenv("CUHE_SCHED_POLICY", std::to_string(policyNo).c_str(), 1);
	sched::setBatchRunner(batches ? batchRunner : nullptr, cap);
	sched::start(workers);
	if (!sched::on()) fail("start() did not switch the mode on");
	if (sched::threads() != workers * ndev) fail("not the number of workers per device that was asked for", sched::threads(), ndev);
	std::mt19937 rng(seed);
	std::vector<sched::Node *> nodes(nodesN);
	std::vector<int> nodeDev(nodesN), lastWrite(nodesN, -1);
	std::vector<std::vector<int>> readers(nodesN);
	for (int i = 0; i < nodesN; ++i) { nodes[i] = sched::newNode(nullptr); nodeDev[i] = (int)(rng() % ndev); }
	for (Gate *g : gates) delete g;
	gates.assign(ngates, nullptr); pendingBySubject.clear();          // (sized up front: the workers index it while the client still records)
	std::vector<sched::Task *> kept;
	for (int t = 0; t < ngates; ++t) {
		Gate *g = new Gate;
		g->id = t;
		const int w = (int)(rng() % nodesN);
		g->dev = nodeDev[w];
		// operands on the same device mostly; sometimes a "moveTo": the written node changes device (recorded on the source device)
		std::vector<int> rd;
		const int nr = (int)(rng() % 3);
		for (int k = 0; k < nr; ++k) { int r = (int)(rng() % nodesN); if (nodeDev[r] == g->dev || rng() % 4 == 0) rd.push_back(r); }
		if (rng() % 7 == 0 && !rd.empty()) rd.push_back(rd[0]);          // an operand listed twice
		const bool isBatchable = rng() % 3 != 0;
		g->kind = isBatchable ? 1 + (int)(rng() % 3) : 0;
		const bool release = blocks && rng() % 5 == 0;                    // the polynomial is reset: a release-only task
		if (release) { rd.clear(); g->kind = sched::kReleaseOnly; }
		else if (!isBatchable && rng() % 3 == 0) { g->kind = sched::kHostBlocking; g->blocking = true; }      // an upload from a host value
		g->w = w; g->rd = rd; g->allocs = blocks && rng() % 2 == 0;
		g->key = (long)(rng() % 2);
		// the dependency rule, restated: after the last writer of everything touched, and after every reader since of what is written
		std::set<int> mf;
		for (int r : rd) if (lastWrite[r] >= 0) mf.insert(lastWrite[r]);
		if (lastWrite[w] >= 0) mf.insert(lastWrite[w]);
		for (int r : readers[w]) mf.insert(r);
		g->mustFollow.assign(mf.begin(), mf.end());
		for (int r : rd) if (r != w) readers[r].push_back(t);
		lastWrite[w] = t; readers[w].clear();
		gates[t] = g;
		std::vector<sched::Node *> reads, writes(1, nodes[w]);
		for (int r : rd) reads.push_back(nodes[r]);
		if (g->kind > 0) { std::lock_guard<std::mutex> lk(gm); pendingBySubject[nodes[w]].push_back(g); }
		const bool keep = rng() % 50 == 0;
		if (keep && g->kind == 0) g->blocking = true;
		sched::Task *task = sched::submit(g->dev, reads, writes, [g](void *s) {
			if (g->kind == sched::kReleaseOnly) { runRelease(g, s); return; }
			if (g->kind > 0) {                                   // ran alone although batchable: take it off its subject's list
				std::lock_guard<std::mutex> lk(gm);
				for (auto &kv : pendingBySubject) { auto &v = kv.second; for (size_t i = 0; i < v.size(); ++i) if (v[i] == g) { v.erase(v.begin() + i); goto done; } }
				done:;
			}
			runGate(g, s);
		}, keep, g->kind, g->key, g->kind > 0 ? nodes[w] : nullptr, nullptr, nullptr);
		if (keep) kept.push_back(task);
		if (!blocks && rng() % 40 == 0) nodeDev[w] = (int)(rng() % ndev);          // the node "moved" (not in the block model: a block stays with its device's cache)
		if (rng() % 97 == 0) sched::waitNode(nodes[w]);
	}
	for (sched::Task *k : kept) sched::wait(k);
	sched::drain();
	if (workers > 1) for (int sid : blockingStreams) if (groupStreams.count(sid)) fail("a host-blocking task ran on a worker that takes groups", sid, workers);
	for (Gate *g : gates) if (g->runs.load() != 1) fail("gate did not run exactly once", g->id, g->runs.load());
	if (sched::stats().tasks != ngates) fail("task count", (int)sched::stats().tasks, ngates);
	for (sched::Node *n : nodes) sched::releaseNode(n);
	for (void *&p : blk) if (p) { sched::forgetBlock(p); free(p); p = nullptr; }      // blocks the nodes still hold
	sched::stop();
	if (sched::on() || sched::threads() != 0) fail("stop() left workers behind");
	blocksOn = false;
}
// Several CLIENT threads, each recording chains of gates on its own polynomials and blocking on a result after every chain (the reference's PRINCE
// client: one OpenMP thread per S-box, ZZX in, gates, x2z -- Prince.cu:188-322).  While a client waits for ONE result the device runs for latency: groups
// go before they are complete although other clients' uploads / copies down are in flight; the rules checked are the same -- host order, device
// order, every gate once, the result is there when wait() returns, host-blocking tasks stay off the group takers.
void clientThreads(int ndev, int T, int rounds, int workers, int policyNo, unsigned seed) {
	mock::numGpus = ndev;
	groupStreams.clear(); blockingStreams.clear();
	blocksOn = false;
	setenv("CUHE_SCHED_POLICY", std::to_string(policyNo).c_str(), 1);
	sched::setBatchRunner(batchRunner, 64);
	sched::start(workers);
	const int per = 12, mine = 6;
	for (Gate *g : gates) delete g;
	gates.assign((size_t)T * rounds * per, nullptr); pendingBySubject.clear();
	std::atomic<int> nextGate{0};
	std::vector<std::thread> clients;
	for (int t = 0; t < T; ++t) clients.emplace_back([&, t] {
		std::mt19937 rng(seed + 31u * t);
		const int dev = t % ndev;
		std::vector<sched::Node *> nodes(mine);
		for (auto &n : nodes) n = sched::newNode(nullptr);
		std::vector<int> lastWrite(mine, -1);
		std::vector<std::vector<int>> readers(mine);
		for (int r = 0; r < rounds; ++r)
			for (int k = 0; k < per; ++k) {
				const bool last = k == per - 1;
				Gate *g = new Gate;
				g->id = nextGate.fetch_add(1); g->dev = dev;
				const int w = (int)(rng() % mine);
				std::vector<int> rd;
				for (int i = (int)(rng() % 3); i > 0; --i) rd.push_back((int)(rng() % mine));
				g->kind = last ? 0 : k < 2 ? (int)sched::kHostBlocking : 1 + (int)(rng() % 3);       // two uploads, gates, the copy down
				g->blocking = last || g->kind == sched::kHostBlocking;
				g->key = (long)(r % 2); g->w = w; g->rd = rd;
				std::set<int> mf;
				for (int x : rd) if (lastWrite[x] >= 0) mf.insert(lastWrite[x]);
				if (lastWrite[w] >= 0) mf.insert(lastWrite[w]);
				for (int x : readers[w]) mf.insert(x);
				g->mustFollow.assign(mf.begin(), mf.end());
				for (int x : rd) if (x != w) readers[x].push_back(g->id);
				lastWrite[w] = g->id; readers[w].clear();
				{ std::lock_guard<std::mutex> lk(gm); gates[g->id] = g; if (g->kind > 0) pendingBySubject[nodes[w]].push_back(g); }
				std::vector<sched::Node *> reads, writes(1, nodes[w]);
				for (int x : rd) reads.push_back(nodes[x]);
				sched::Task *task = sched::submit(dev, reads, writes, [g](void *s) {
					if (g->kind > 0) {
						std::lock_guard<std::mutex> lk(gm);
						for (auto &kv : pendingBySubject) { auto &v = kv.second; for (size_t i = 0; i < v.size(); ++i) if (v[i] == g) { v.erase(v.begin() + i); goto done; } }
						done:;
					}
					runGate(g, s);
				}, last, g->kind, g->key, g->kind > 0 ? nodes[w] : nullptr, nullptr, nullptr);
				if (last) { sched::wait(task); if (g->runs.load() != 1) fail("wait() returned before the task had run", g->id); }
			}
		for (auto &n : nodes) sched::releaseNode(n);
	});
	for (auto &c : clients) c.join();
	sched::drain();
	if (workers > 1) for (int sid : blockingStreams) if (groupStreams.count(sid)) fail("a host-blocking task ran on a worker that takes groups", sid, workers);
	for (Gate *g : gates) if (!g || g->runs.load() != 1) fail("gate did not run exactly once (client threads)", g ? g->id : -1);
	if (sched::stats().tasks != (long)gates.size()) fail("task count (client threads)", (int)sched::stats().tasks, (int)gates.size());
	sched::stop();
}
//  blocks and threads that have no stream on the block's device.  A task of device 0 that allocates on device 1 (work after a moveTo /
// copyTo) cannot be ordered behind the last use of a block in device 1's cache: it must get a FRESH block from the library, and handing it
// back must leave no size entry behind -- the library may issue the same address again with another size, and the block must then be cached
// under ITS size.
void crossDeviceBlocks() {
	mock::numGpus = 2;
	sched::setBatchRunner(nullptr, 1);
	sched::start(2);
	sched::Node *n0 = sched::newNode(nullptr), *n1 = sched::newNode(nullptr);
	void *cached = nullptr, *foreign = nullptr, *again = nullptr, *small = nullptr;
	auto on = [&](int dev, sched::Node *n, std::function<void()> f) { sched::wait(sched::submit(dev, {}, std::vector<sched::Node *>(1, n), [f](void *s) { mock::launch(s); f(); }, true)); };
	on(1, n1, [&] { cached = sched::taskAlloc(1, 4096); if (!sched::taskFree(1, cached)) fail("a worker of device 1 could not hand its own block back"); });
	const long m0 = mock::mallocs.load();
	on(0, n0, [&] {
		foreign = sched::taskAlloc(1, 4096);                      // a worker of device 0: no stream on device 1
		if (foreign == cached) fail("a thread without a stream on the device got a cached block (nothing orders it behind the block's last use)");
		if (mock::mallocs.load() != m0 + 1) fail("the cross-device allocation did not come from the library");
		if (sched::taskFree(1, foreign)) fail("taskFree kept a block although the caller has no stream on that device");
		mock::recycle = true; cuhe_hip_free(1, foreign);          // what devFree does when taskFree says no; the mock re-issues this address next
	});
	on(1, n1, [&] {
		if (sched::taskAlloc(1, 4096) != cached) fail("the cached block of device 1 is gone");
		again = sched::taskAlloc(1, 8192);                        // the library hands the old address out with another size
		if (again != foreign) fail("mock did not recycle the address");
		if (!sched::taskFree(1, again)) fail("taskFree refused a block taskAlloc handed out");
		small = sched::taskAlloc(1, 4096);
		if (small == again) fail("a block was cached under the size a stale entry remembered: an 8192-byte block came back for 4096");
		if (sched::taskAlloc(1, 8192) != again) fail("the re-issued block is not cached under its own size");
	});
	sched::drain();
	sched::releaseNode(n0); sched::releaseNode(n1);
	sched::stop();
}
}  // namespace

// ---------------------------------------------------------------- the gate program of a homomorphic PRINCE block, recorded the way
// tests/cxx/test_prince_flow.cpp records it (one client thread, S-box by S-box: 24 579 gates), with the host cost of issuing a gate
// emulated by a spin: how the batch policies group the gates, without a GPU.  `prince <policy> <workers> [devices]`
namespace princesim {
enum { X2C = 1, X2N = 2, RELIN = 3, MS = 4, AND = 5, XOR = 6, COPY = 7, NOT = 8 };
struct Ct { sched::Node *n = nullptr; int level = 0, domain = 2; bool prod = false; int dev = 0; };
long key(int level, int domain, bool prod) { return (long)level | (long)domain << 8 | (long)(prod ? 1 : 0) << 12; }
std::atomic<long> lone{0};
void spin(double us) { static const bool off = getenv("SIM_NOSPIN") != nullptr; if (off) return; const auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {} }
void simBatch(int, sched::Node *const *, sched::Node *const *, sched::Node *const *, int count, void *stream) { mock::launch(stream); spin(25.0 + 0.4 * count); }
void gate(int kind, Ct &out, std::vector<Ct *> reads, long k) {
	std::vector<sched::Node *> r; for (Ct *c : reads) if (c->n != out.n) r.push_back(c->n);
	sched::submit(out.dev, r, std::vector<sched::Node *>(1, out.n), [](void *s) { mock::launch(s); ++lone; spin(8.0); }, false, kind, k, out.n, nullptr, nullptr);
}
void fresh(Ct &c, int dev) { if (c.n) sched::releaseNode(c.n); c = Ct(); c.n = sched::newNode(nullptr); c.dev = dev; }
void x2n(Ct &c) { if (c.domain == 3) return; gate(X2N, c, {}, key(c.level, 2, false)); c.domain = 3; }
void x2c(Ct &c) { if (c.domain == 2) return; gate(X2C, c, {}, key(c.level, 3, c.prod)); c.domain = 2; c.prod = false; }
void cAnd(Ct &o, Ct &a, Ct &b) { fresh(o, a.dev); gate(AND, o, {&a, &b}, key(a.level, 3, false)); o.level = a.level; o.domain = 3; o.prod = true; }
void relin(Ct &c) { gate(RELIN, c, {}, key(c.level, c.domain, c.prod)); c.domain = 2; c.prod = false; }
void modSwitch(Ct &c) { gate(MS, c, {}, key(c.level, c.domain, c.prod)); c.domain = 2; c.prod = false; ++c.level; }
void copy(Ct &o, Ct &a) { fresh(o, a.dev); gate(COPY, o, {&a}, key(a.level, a.domain, false)); o.level = a.level; o.domain = a.domain; o.prod = a.prod; }
void cXor(Ct &o, Ct &a, Ct &b) { gate(XOR, o, {&a, &b}, key(a.level, a.domain, false)); }
void cNot(Ct &o) { gate(NOT, o, {&o}, key(o.level, 2, false)); }
void accumulate(Ct &out, bool &has, Ct &term) { if (!has) { copy(out, term); has = true; } else cXor(out, out, term); }
void sbox(Ct *s[4], std::mt19937 &rng) {
	Ct &a = *s[0], &b = *s[1], &c = *s[2], &d = *s[3];
	x2n(a); x2n(b); x2n(c); x2n(d);
	Ct ab, ac, ad, bc, bd, cd;
	cAnd(ab, a, b); cAnd(ac, a, c); cAnd(ad, a, d); cAnd(bc, b, c); cAnd(bd, b, d); cAnd(cd, c, d);
	relin(ab); relin(cd);
	Ct *lvl1[10] = {&ab, &ac, &ad, &bc, &bd, &cd, &a, &b, &c, &d};
	for (Ct *x : lvl1) modSwitch(*x);
	Ct out[4]; bool has[4] = {false, false, false, false};
	for (int o = 0; o < 4; ++o) { int n = 0; for (Ct *t : lvl1) if (rng() % 2 || (t == lvl1[9] && !n)) { accumulate(out[o], has[o], *t); ++n; } }
	x2n(a); x2n(b); x2n(c); x2n(d); x2n(ab); x2n(cd);
	Ct abd, acd, bcd, abc;
	cAnd(abd, ab, d); cAnd(acd, cd, a); cAnd(bcd, cd, b); cAnd(abc, ab, c);
	x2c(abd); x2c(acd); x2c(bcd); x2c(abc);
	Ct *high[4] = {&abd, &acd, &bcd, &abc};
	for (int o = 0; o < 4; ++o) {
		for (Ct *t : high) if (rng() % 2) accumulate(out[o], has[o], *t);
		if (rng() % 2) cNot(out[o]);
		relin(out[o]); modSwitch(out[o]);
	}
	Ct *all[] = {&ab, &ac, &ad, &bc, &bd, &cd, &abd, &acd, &bcd, &abc, &a, &b, &c, &d};
	for (Ct *x : all) { sched::releaseNode(x->n); x->n = nullptr; }
	for (int o = 0; o < 4; ++o) *s[o] = out[o];
}
int run(int pol, int workers, int ndev) {
	mock::numGpus = ndev;
	setenv("CUHE_SCHED_POLICY", std::to_string(pol).c_str(), 1);
	setenv("CUHE_SCHED_STATS", "1", 1); setenv("CUHE_SCHED_TRACE", "1", 1);
	sched::setBatchRunner(simBatch, 128);
	sched::start(workers);
	std::mt19937 rng(7);
	std::vector<Ct> state(64), k1(64), k0(64);
	for (int i = 0; i < 64; ++i) { fresh(state[i], 0); fresh(k1[i], 0); fresh(k0[i], 0); }
	// hold every worker until the circuit has been recorded (one blocking gate per worker): what is compared is how the policies
	// group a KNOWN program, not how far the client happens to be ahead of the workers on a loaded test machine
	std::atomic<bool> go{false};
	std::vector<Ct> holders(workers * ndev);
	for (int i = 0; i < workers * ndev; ++i) {
		fresh(holders[i], i % ndev);
		sched::submit(holders[i].dev, {}, std::vector<sched::Node *>(1, holders[i].n), [&go](void *s) { mock::launch(s); while (!go.load()) std::this_thread::yield(); });
	}
	const auto t0 = std::chrono::steady_clock::now();
	auto addKey = [&](std::vector<Ct> &k) { for (int i = 0; i < 64; ++i) cXor(state[i], state[i], k[i]); };
	auto addConstant = [&]() { for (int i = 0; i < 64; ++i) if (rng() % 2) cNot(state[i]); };
	auto mPrime = [&]() {
		std::vector<Ct> next(64);
		for (int i = 0; i < 64; ++i) { copy(next[i], state[(i * 7 + 1) % 64]); cXor(next[i], next[i], state[(i * 11 + 3) % 64]); cXor(next[i], next[i], state[(i * 13 + 5) % 64]); }
		for (int i = 0; i < 64; ++i) { sched::releaseNode(state[i].n); state[i] = next[i]; }
	};
	auto sboxLayer = [&]() {
		for (int i = 0; i < 16; ++i) {
			Ct *s[4] = {&state[4 * i], &state[4 * i + 1], &state[4 * i + 2], &state[4 * i + 3]};
			const int dev = i % ndev;
			for (Ct *x : s) if (x->dev != dev) { gate(0, *x, {}, 0); x->dev = dev; }           // moveTo (recorded on the source device)
			sbox(s, rng);
			for (Ct *x : s) if (x->dev != 0) { gate(0, *x, {}, 0); x->dev = 0; }
		}
		for (int i = 0; i < 64; ++i) { modSwitch(k1[i]); modSwitch(k1[i]); }
	};
	addKey(k0); addKey(k1); addConstant();
	for (int r = 0; r < 12; ++r) { sboxLayer(); mPrime(); addConstant(); addKey(k1); }
	const auto t1 = std::chrono::steady_clock::now();
	go.store(true);
	sched::drain();
	const auto t2 = std::chrono::steady_clock::now();
	printf("policy %d, %d workers per device, %d device(s): recorded in %.3f s, issued after %.3f s (host side only; %ld gates ran alone)\n", pol, workers, ndev,
	       std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t0).count(), lone.load());
	for (auto *v : {&state, &k1, &k0, &holders}) for (Ct &c : *v) sched::releaseNode(c.n);
	sched::stop();
	return 0;
}
}  // namespace princesim

int main(int argc, char **argv) {
	if (argc > 1 && std::string(argv[1]) == "prince") return princesim::run(argc > 2 ? atoi(argv[2]) : 1, argc > 3 ? atoi(argv[3]) : 3, argc > 4 ? atoi(argv[4]) : 1);
	const int rounds = argc > 1 ? atoi(argv[1]) : 3;
	long programs = 0;
	for (int round = 0; round < rounds; ++round)
		for (int ndev : {1, 2, 8})
			for (int pol : {0, 1, 2})
				for (int batches = 0; batches < 2; ++batches) {
					const long b0 = batchesSeen.load();
					program(ndev, 40 + 10 * ndev, 3000, 1000u * round + 100u * ndev + 10u * pol + batches, batches != 0, pol, batches ? (round % 2 ? 128 : 5) : 1);
					if (batches && batchesSeen.load() == b0) fail("no batch was formed", ndev, pol);
					if (!batches && batchesSeen.load() != b0) fail("the batch runner ran although it was switched off");
					++programs;
					if (ndev <= 2) {          // the same program with device blocks taken, used and released inside the tasks
						program(ndev, 40 + 10 * ndev, 3000, 7000u * round + 100u * ndev + 10u * pol + batches, batches != 0, pol, batches ? 64 : 1, true);
						++programs;
						// ... and with ONE and TWO workers per device (CUHE_SCHED_THREADS=1: the single worker runs the regular tasks AND takes the groups)
						for (int workers : {1, 2}) {
							program(ndev, 40 + 10 * ndev, 1500, 9000u * round + 100u * ndev + 10u * pol + batches + 1000u * workers, batches != 0, pol, batches ? 32 : 1, workers == 1, workers);
							++programs;
						}
					}
				}
	for (int round = 0; round < rounds; ++round)
		for (int ndev : {1, 2})
			for (int workers : {1, 2, 3})
				for (int pol : {1, 2}) { clientThreads(ndev, 8, 40, workers, pol, 500u * round + 10u * workers + pol); ++programs; }
	crossDeviceBlocks();
	if (blocksTaken.load() < 1000 || blocksReused.load() * 4 < blocksTaken.load() || blocksReleased.load() * 2 < blocksTaken.load())
		fail("the block model did not exercise reuse", (int)blocksTaken.load(), (int)blocksReused.load());
	printf("blocks: %ld taken inside tasks, %ld of them had been used before, %ld released by release-only tasks\n", blocksTaken.load(), blocksReused.load(), blocksReleased.load());
	if (maxBatchSeen.load() > 128) fail("batch above the cap", maxBatchSeen.load());
	printf("%ld programs, %ld batches of %ld gates (largest %d), %ld event records, %ld stream waits, %ld mock streams\n", programs, batchesSeen.load(), batchedGates.load(),
	       maxBatchSeen.load(), mock::eventRecords.load(), mock::streamWaits.load(), (long)mock::streams.size());
	printf(failures.load() ? "FAILED (%ld)\n" : "ALL PASSED\n", failures.load());
	return failures.load() ? 1 : 0;
}
