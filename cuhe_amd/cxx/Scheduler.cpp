// Scheduler.cpp -- task graph and worker threads of the gate scheduler (see Scheduler.h).
//
// Ordering on the GPU.  A task runs on the stream of the worker that picked it; a task it depends on may have run on another
// worker's stream.  hipEventRecord is the expensive call here (5 us of host time with four threads launching, 18 us with
// eight: tools/ubench_launch.hip, profiles/r04_sched_prince.txt), so no task records an event of its own.  Every worker
// stream counts the tasks it has issued (`seq`) and owns ONE event; a consumer on another stream that needs "task #k of
// that stream has finished" records that event on the producer's stream only if its last record does not cover #k yet --
// the record lands behind #k (and possibly behind later tasks: more ordering than asked for, never less) -- and waits for
// it.  With the per-worker queues below most dependencies stay on one stream and need nothing at all.
#include "Scheduler.h"
#include "CuHE.h"
#include "Debug.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace cuHE {
namespace sched {

// one worker's stream on one device
struct StreamState {
	void *stream = nullptr, *event = nullptr;
	int dev = 0;
	std::atomic<long> seq{0};              // tasks issued on this stream so far (written by its worker only)
	std::mutex m;                          // guards `event` / `covered`
	long covered = 0;                      // the event's last record lies behind task #covered
};
struct Task {
	std::function<void(void *)> fn;
	int dev = 0;
	std::vector<Task *> deps;              // every task this one is ordered after (references held until it has run)
	std::vector<Task *> succ;              // tasks that wait for this one to be issued
	std::vector<Node *> nodes;             // references held until it has run
	int pending = 0;                       // dependencies not issued yet
	bool issued = false;
	StreamState *ss = nullptr; long seq = 0;     // where it ran: task #seq of that stream
	int kind = 0; long key = 0; Node *subject = nullptr, *op1 = nullptr, *op2 = nullptr;      // batchable gate (Scheduler.h)
	int refs = 1;                          // the graph itself until the task has run; + nodes, successors, waiters
};

namespace {
std::mutex mu;                             // guards the whole graph: tasks are ~25k per PRINCE block, each touched a few times
std::condition_variable cvReady, cvDone;
std::deque<Task *> ready;                  // tasks that were ready when the client recorded them
std::vector<std::deque<Task *>> local;     // per worker: tasks its own tasks made ready (newest at the back; thieves take the oldest)
int stealing = 1;                          // CUHE_SCHED_LOCAL=0: one shared queue
// batchable ready tasks by (kind, key, device); they run when the workers have no other ready task (the gates that FEED a
// group -- the products before the relinearisations of a layer -- are issued first, so that the group is as large as the circuit allows)
struct GroupKey { int kind, dev; long key; bool operator<(const GroupKey &o) const { return kind != o.kind ? kind < o.kind : dev != o.dev ? dev < o.dev : key < o.key; } };
std::map<GroupKey, std::deque<Task *>> staged;
long stagedCount = 0, batchesRun = 0, batchedTasks = 0;
int busyRegular = 0;                       // workers inside a non-batch task: more members of a group may still appear
BatchRunner batchRunner = nullptr; int maxBatch = 1;
std::vector<std::thread> workers;
bool active = false, stopping = false;
long outstanding = 0, totalTasks = 0, totalWaits = 0, totalRecords = 0, maxQueued = 0;
std::vector<char> devUsed;                      // devices some task has run on
double busySeconds = 0, idleSeconds = 0, gateSeconds = 0, orderSeconds = 0;       // summed over the workers (CUHE_SCHED_STATS=1 prints them at stop())
int startedWorkers = 0;
typedef std::chrono::steady_clock clk;
thread_local bool tlsWorker = false;
thread_local void *tlsStream = nullptr;
thread_local std::vector<StreamState *> tlsStreams;    // this worker's stream per device (never freed: tasks point at them)

void unrefTask(Task *t) {                  // mu held
	if (--t->refs > 0) return;
	delete t;
}
// mu held; objects whose node died are handed back to be deleted outside the lock (their destructors call into the library)
void unrefNode(Node *n, std::vector<CuPolynomial *> &dead) {
	if (--n->refs > 0) return;
	if (n->lastWrite) unrefTask(n->lastWrite);
	for (Task *r : n->readers) unrefTask(r);
	if (n->obj) dead.push_back(n->obj);
	delete n;
}
// readers of a node that is never written (a key bit read by every round) would pile up: of the readers that have been
// issued, the last one per stream stands for the earlier ones of that stream
void pruneReaders(Node *n) {               // mu held
	const std::vector<Task *> all = n->readers;
	std::vector<char> shadowed(all.size(), 0);
	for (size_t i = 0; i < all.size(); ++i) {
		Task *r = all[i];
		if (!r->issued) continue;
		for (size_t j = 0; j < all.size() && !shadowed[i]; ++j) {
			Task *o = all[j];
			shadowed[i] = j != i && o->issued && o->ss == r->ss && (o->seq > r->seq || (o->seq == r->seq && j > i));
		}
	}
	// one reference per ENTRY (submit() enters a task once per node, but the count must not depend on that: ADVICE r04)
	n->readers.clear();
	for (size_t i = 0; i < all.size(); ++i) if (!shadowed[i]) n->readers.push_back(all[i]);
	for (size_t i = 0; i < all.size(); ++i) if (shadowed[i]) unrefTask(all[i]);
}
// stream `s` (of device `dev`) waits until task #seq of `from` has finished; returns the number of events recorded (0 or 1)
int orderAfter(int dev, void *s, StreamState *from, long seq) {
	std::lock_guard<std::mutex> lk(from->m);
	int recorded = 0;
	if (from->covered < seq) {
		const long now = from->seq.load(std::memory_order_acquire);    // everything up to #now has been enqueued
		CSC(cuhe_hip_event_record(from->dev, from->event, from->stream));
		from->covered = now; recorded = 1;
	}
	CSC(cuhe_hip_stream_wait_event(dev, s, from->event));
	return recorded;
}

// ---- device blocks released and taken inside tasks.  The library's stream-ordered pool (cuhe_hip_malloc_stream) hands a
// block freed on one stream to another stream only behind everything enqueued on the first; here a released block carries
// "task #k of stream S" -- its last use -- and the taker is ordered behind exactly that, usually an event record of long ago.
struct Block { void *ptr; StreamState *ss; long seq; };
std::mutex cacheMu;
std::vector<std::unordered_map<size_t, std::deque<Block>>> cache;       // per device, by size
std::unordered_map<void *, size_t> cacheSize;                           // blocks handed out by taskAlloc
long cacheHits = 0, cacheForeign = 0, cacheMisses = 0;
unsigned long long cacheGeneration = 0;    // cuhe_hip_generation() the cached blocks belong to
void cacheCheckGeneration() {              // cacheMu held: a cuhe_hip_shutdown since took every block with it
	const unsigned long long g = cuhe_hip_generation();
	if (g != cacheGeneration) { cache.clear(); cacheSize.clear(); cacheGeneration = g; }
}
void flushCache() {                        // the device has been synchronised: everything goes back to the library's pool
	std::lock_guard<std::mutex> lk(cacheMu);
	cacheCheckGeneration();
	for (size_t d = 0; d < cache.size(); ++d)
		for (auto &bySize : cache[d])
			for (Block &b : bySize.second) { cacheSize.erase(b.ptr); CSC(cuhe_hip_free((int)d, b.ptr)); }
	cache.clear();
}

// mu held: this worker's newest task, else the oldest the client recorded, else the oldest of the fullest other worker
Task *takeTask(int me) {
	if (!local[me].empty()) { Task *t = local[me].back(); local[me].pop_back(); return t; }
	if (!ready.empty()) { Task *t = ready.front(); ready.pop_front(); return t; }
	size_t best = 0; int from = -1;
	for (size_t w = 0; w < local.size(); ++w) if (local[w].size() > best) { best = local[w].size(); from = (int)w; }
	if (from < 0) return nullptr;
	Task *t = local[from].front(); local[from].pop_front();
	return t;
}
// mu held: the fullest staged group, up to maxBatch of its tasks (oldest first) -- only when it is full or no worker is inside a
// regular task any more (what such a task makes ready may belong to the group)
bool takeBatch(std::vector<Task *> &batch) {
	if (stagedCount == 0) return false;
	auto best = staged.end();
	for (auto it = staged.begin(); it != staged.end(); ++it) if (best == staged.end() || it->second.size() > best->second.size()) best = it;
	if (best == staged.end() || best->second.empty()) return false;
	if ((int)best->second.size() < maxBatch && busyRegular > 0) return false;
	while (!best->second.empty() && (int)batch.size() < maxBatch) { batch.push_back(best->second.front()); best->second.pop_front(); --stagedCount; }
	if (best->second.empty()) staged.erase(best);
	return true;
}
// mu held: a task whose dependencies have all been issued
void makeReady(Task *t, int me, Task **next) {
	if (t->kind && batchRunner && maxBatch > 1) {
		staged[GroupKey{t->kind, t->dev, t->key}].push_back(t); ++stagedCount;
		cvReady.notify_one();
		return;
	}
	if (next && !*next) { *next = t; return; }            // follow the chain on this stream: no event wait, warm scratch
	if (me >= 0 && stealing) local[me].push_back(t); else ready.push_back(t);
	cvReady.notify_one();
}
// streams of workers that have gone (setScheduled(false) ; setScheduled(true) cycles): a stream costs ~10 ms to create, and issued
// tasks keep pointing at their StreamState, so the states are never destroyed -- the next workers take them over
std::mutex idleMu;
std::vector<std::vector<StreamState *>> idleStreams;     // per device
StreamState *streamOf(int dev) {
	if ((int)tlsStreams.size() <= dev) tlsStreams.resize(dev + 1, nullptr);
	if (!tlsStreams[dev]) {
		StreamState *ns = nullptr;
		{
			std::lock_guard<std::mutex> lk(idleMu);
			if ((int)idleStreams.size() > dev && !idleStreams[dev].empty()) { ns = idleStreams[dev].back(); idleStreams[dev].pop_back(); }
		}
		if (!ns) {
			ns = new StreamState;
			ns->dev = dev;
			CSC(cuhe_hip_stream_create(dev, &ns->stream));
			CSC(cuhe_hip_event_create(dev, &ns->event));
		}
		tlsStreams[dev] = ns;
	}
	return tlsStreams[dev];
}
struct ReturnStreams { ~ReturnStreams() {                 // at worker exit
	std::lock_guard<std::mutex> lk(idleMu);
	for (size_t d = 0; d < tlsStreams.size(); ++d) if (tlsStreams[d]) { if (idleStreams.size() <= d) idleStreams.resize(d + 1); idleStreams[d].push_back(tlsStreams[d]); }
	tlsStreams.clear();
} };
void workerMain(int me) {
	tlsWorker = true;
	ReturnStreams giveBack;
	// a stream costs ~10 ms to create: before the first task, not inside it
	if (cuhe_hip_is_initialised()) for (int d = 0; d < cuhe_hip_num_gpus(); ++d) streamOf(d);
	std::unique_lock<std::mutex> lk(mu);
	++startedWorkers; cvDone.notify_all();
	Task *next = nullptr;
	std::vector<Task *> batch;
	for (;;) {
		batch.clear();
		if (next) { batch.push_back(next); next = nullptr; }
		else {
			const auto w0 = clk::now();
			for (;;) {
				if (Task *t = takeTask(me)) { batch.push_back(t); break; }
				if (takeBatch(batch)) break;
				if (stopping) return;
				cvReady.wait(lk);
			}
			idleSeconds += std::chrono::duration<double>(clk::now() - w0).count();
		}
		const auto b0 = clk::now();
		const bool regular = batch.size() == 1 && !(batch[0]->kind && batchRunner && maxBatch > 1);
		if (regular) ++busyRegular;
		const int dev = batch[0]->dev;
		if ((int)devUsed.size() <= dev) devUsed.resize(dev + 1, 0);
		devUsed[dev] = 1;
		lk.unlock();
		StreamState *ss = streamOf(dev);
		void *s = ss->stream;
		long waits = 0, records = 0;
		// one wait per foreign stream: behind the latest of the dependencies that ran there
		std::vector<std::pair<StreamState *, long>> latest;
		for (Task *t : batch)
			for (Task *d : t->deps) {
				if (d->ss == ss) continue;          // same stream: already ordered
				bool found = false;
				for (auto &e : latest) if (e.first == d->ss) { if (d->seq > e.second) e.second = d->seq; found = true; }
				if (!found) latest.push_back({d->ss, d->seq});
			}
		for (auto &e : latest) { records += orderAfter(dev, s, e.first, e.second); ++waits; }
		tlsStream = s;
		const auto f0 = clk::now();
		if (batch.size() == 1) batch[0]->fn(s);
		else {
			std::vector<Node *> subjects, o1, o2;
			for (Task *t : batch) { subjects.push_back(t->subject); o1.push_back(t->op1); o2.push_back(t->op2); }
			batchRunner(batch[0]->kind, subjects.data(), o1.data(), o2.data(), (int)subjects.size(), s);
		}
		for (Task *t : batch) t->fn = nullptr;      // the closures' captures go before the graph lock is taken again
		const auto f1 = clk::now();
		std::vector<CuPolynomial *> dead;
		const long seq = ss->seq.load(std::memory_order_relaxed) + 1;
		ss->seq.store(seq, std::memory_order_release);
		lk.lock();
		if (regular) --busyRegular;
		if (batch.size() > 1) { ++batchesRun; batchedTasks += (long)batch.size(); }
		totalWaits += waits; totalRecords += records;
		busySeconds += std::chrono::duration<double>(clk::now() - b0).count();
		gateSeconds += std::chrono::duration<double>(f1 - f0).count();
		orderSeconds += std::chrono::duration<double>(f0 - b0).count();
		for (Task *t : batch) {
			t->ss = ss; t->seq = seq; t->issued = true;
			for (Task *d : t->deps) unrefTask(d);
			t->deps.clear();
			for (Node *n : t->nodes) unrefNode(n, dead);
			t->nodes.clear();
			for (Task *x : t->succ) if (--x->pending == 0) makeReady(x, me, &next);
			t->succ.clear();
			--outstanding;
			unrefTask(t);
		}
		if (stagedCount && busyRegular == 0) cvReady.notify_all();     // a group that waited for the regular work to drain
		cvDone.notify_all();
		if (!dead.empty()) {
			lk.unlock();
			for (CuPolynomial *p : dead) delete p;
			lk.lock();
		}
	}
}

struct AtExit { ~AtExit() {                 // idle workers must not outlive the process's static state
	if (tlsWorker) { for (auto &w : workers) w.detach(); return; }      // exit(-1) from a failed call inside a task: nothing to wait for
	{ std::lock_guard<std::mutex> lk(mu); stopping = true; }
	cvReady.notify_all();
	for (auto &w : workers) if (w.joinable()) w.join();
	workers.clear();
} } atExit;
} // namespace

bool on() { return active; }
bool inWorker() { return tlsWorker; }
void *workerStream() { return tlsStream; }
int threads() { return (int)workers.size(); }

void start(int n) {
	std::unique_lock<std::mutex> lk(mu);
	if (active) return;
	if (n <= 0) { const char *e = getenv("CUHE_SCHED_THREADS"); n = e ? atoi(e) : 0; }
	if (n <= 0) n = 3;                          // PRINCE gate by gate: 0.114-0.121 s with 3 workers, 0.126-0.133 with 4 (ready gates in batches of up to 64; profiles/r04_sched_prince.txt)
	stopping = false; startedWorkers = 0;
	if (getenv("CUHE_SCHED_LOCAL")) stealing = atoi(getenv("CUHE_SCHED_LOCAL"));
	local.assign(n, std::deque<Task *>());
	for (int i = 0; i < n; ++i) workers.emplace_back(workerMain, i);
	cvDone.wait(lk, [n] { return startedWorkers == n; });      // their streams exist
	active = true;
}
void drain() {
	std::vector<char> used;
	{
		std::unique_lock<std::mutex> lk(mu);
		cvDone.wait(lk, [] { return outstanding == 0; });
		used = devUsed;
	}
	for (size_t d = 0; d < used.size(); ++d) if (used[d]) CSC(cuhe_hip_device_sync((int)d));
	flushCache();
}
void *taskAlloc(int dev, size_t bytes) {
	StreamState *me = (int)tlsStreams.size() > dev ? tlsStreams[dev] : nullptr;
	Block b{nullptr, nullptr, 0};
	{
		std::lock_guard<std::mutex> lk(cacheMu);
		cacheCheckGeneration();
		if ((int)cache.size() <= dev) cache.resize(dev + 1);
		auto it = cache[dev].find(bytes);
		if (it != cache[dev].end() && !it->second.empty()) {
			std::deque<Block> &q = it->second;
			size_t pick = q.size();
			for (size_t i = q.size(); i-- > 0 && q.size() - i <= 8;) if (q[i].ss == me) { pick = i; break; }     // one of this stream's own: nothing to wait for
			if (pick == q.size()) { pick = 0; ++cacheForeign; } else ++cacheHits;                              // else the one released longest ago
			b = q[pick]; q.erase(q.begin() + pick);
		} else ++cacheMisses;
	}
	if (b.ptr) {
		if (b.ss != me && me) orderAfter(dev, me->stream, b.ss, b.seq);
		return b.ptr;
	}
	void *p = cuhe_hip_malloc(dev, bytes);
	if (!p) return nullptr;
	std::lock_guard<std::mutex> lk(cacheMu);
	cacheSize[p] = bytes;
	return p;
}
void forgetBlock(void *p) {                // released outside a task (the client thread, after a detach): the library owns it again
	std::lock_guard<std::mutex> lk(cacheMu);
	if (!cacheSize.empty()) cacheSize.erase(p);
}
bool taskFree(int dev, void *p) {
	StreamState *me = (int)tlsStreams.size() > dev ? tlsStreams[dev] : nullptr;
	std::lock_guard<std::mutex> lk(cacheMu);
	cacheCheckGeneration();
	auto it = cacheSize.find(p);
	if (it == cacheSize.end() || !me) return false;                 // not one of ours (allocated before the object was attached)
	if ((int)cache.size() <= dev) cache.resize(dev + 1);
	cache[dev][it->second].push_back(Block{p, me, me->seq.load(std::memory_order_relaxed) + 1});     // last use: the running task
	return true;
}
void stop() {
	if (!active) return;
	drain();
	{ std::lock_guard<std::mutex> lk(mu); stopping = true; active = false; }
	cvReady.notify_all();
	for (auto &w : workers) w.join();       // (their streams stay with the library: blocks parked on them settle as they go idle)
	if (getenv("CUHE_SCHED_STATS")) {
		long long ac[4] = {0, 0, 0, 0};
		cuhe_hip_alloc_counters(ac);
		printf("allocator: %lld hipMalloc, %lld pool hits, %lld stream hits, %lld cross-stream hand-overs\n", ac[0], ac[1], ac[2], ac[3]);
		printf("task blocks: %ld from the same stream, %ld from another stream (ordered behind their last use), %ld from the library\n", cacheHits, cacheForeign, cacheMisses);
		printf("batches: %ld calls of the batch runner for %ld gates\n", batchesRun, batchedTasks);
		printf("scheduler: %ld tasks, %ld cross-stream waits on %ld event records, at most %ld tasks recorded ahead; %zu workers busy %.3f s (%.3f in the gates, %.3f ordering streams), idle %.3f s in total\n",
		       totalTasks, totalWaits, totalRecords, maxQueued, workers.size(), busySeconds, gateSeconds, orderSeconds, idleSeconds);
	}
	workers.clear();
}

Node *newNode(CuPolynomial *obj) { Node *n = new Node; n->obj = obj; return n; }
void releaseNode(Node *n) {
	std::vector<CuPolynomial *> dead;
	{ std::lock_guard<std::mutex> lk(mu); unrefNode(n, dead); }
	for (CuPolynomial *p : dead) delete p;
}

void setBatchRunner(BatchRunner r, int mb) {
	std::lock_guard<std::mutex> lk(mu);
	const char *e = getenv("CUHE_SCHED_BATCH");
	batchRunner = (e && atoi(e) == 0) ? nullptr : r;
	maxBatch = e && atoi(e) > 1 ? atoi(e) : mb;
	if (maxBatch > mb) maxBatch = mb;
}
Task *submit(int dev, const std::vector<Node *> &reads, const std::vector<Node *> &writes, std::function<void(void *)> fn, bool keep, int kind, long key, Node *subject,
             Node *op1, Node *op2) {
	if (dev < 0) dev = 0;                       // (an object that was never placed: its task only releases host state)
	Task *t = new Task;
	t->fn = std::move(fn); t->dev = dev; t->kind = kind; t->key = key; t->subject = subject; t->op1 = op1; t->op2 = op2;
	std::lock_guard<std::mutex> lk(mu);
	auto after = [&](Task *d) {
		if (!d) return;
		for (Task *e : t->deps) if (e == d) return;
		++d->refs; t->deps.push_back(d);
		if (!d->issued) { d->succ.push_back(t); ++t->pending; }
	};
	auto written = [&](Node *n) { return std::find(writes.begin(), writes.end(), n) != writes.end(); };
	auto held = [&](Node *n) { return std::find(t->nodes.begin(), t->nodes.end(), n) != t->nodes.end(); };
	for (Node *r : reads) after(r->lastWrite);
	for (Node *w : writes) { after(w->lastWrite); for (Task *r : w->readers) after(r); }
	for (Node *r : reads) {
		if (written(r)) continue;
		if (!r->readers.empty() && r->readers.back() == t) continue;      // listed twice (cAnd(out, x, x)): one entry, one reference
		if (r->readers.size() >= 24) pruneReaders(r);
		r->readers.push_back(t); ++t->refs;
	}
	for (Node *w : writes) {
		if (held(w)) continue;                  // (listed twice)
		for (Task *r : w->readers) unrefTask(r);
		w->readers.clear();
		if (w->lastWrite) unrefTask(w->lastWrite);
		w->lastWrite = t; ++t->refs;
		++w->refs; t->nodes.push_back(w);
	}
	for (Node *r : reads) if (!held(r)) { ++r->refs; t->nodes.push_back(r); }
	if (keep) ++t->refs;
	++outstanding; ++totalTasks;
	if (outstanding > maxQueued) maxQueued = outstanding;
	if (t->pending == 0) makeReady(t, -1, nullptr);
	return t;
}

// the client needs the task's device work finished: an event of its own behind the task on the task's stream
void wait(Task *t) {
	StreamState *ss;
	{
		std::unique_lock<std::mutex> lk(mu);
		cvDone.wait(lk, [t] { return t->issued; });
		ss = t->ss;
	}
	thread_local std::vector<void *> mine;      // per client thread and device
	if ((int)mine.size() <= ss->dev) mine.resize(ss->dev + 1, nullptr);
	if (!mine[ss->dev]) CSC(cuhe_hip_event_create(ss->dev, &mine[ss->dev]));
	CSC(cuhe_hip_event_record(ss->dev, mine[ss->dev], ss->stream));
	CSC(cuhe_hip_event_sync(ss->dev, mine[ss->dev]));
	std::lock_guard<std::mutex> lk(mu);
	unrefTask(t);
}
void waitNode(Node *n) {
	std::vector<Task *> ts;
	{
		std::lock_guard<std::mutex> lk(mu);
		if (n->lastWrite) { ++n->lastWrite->refs; ts.push_back(n->lastWrite); }
		for (Task *r : n->readers) { ++r->refs; ts.push_back(r); }
	}
	for (Task *t : ts) wait(t);
}
Stats stats() { std::lock_guard<std::mutex> lk(mu); return Stats{totalTasks, totalWaits, maxQueued}; }

} // namespace sched
} // namespace cuHE
