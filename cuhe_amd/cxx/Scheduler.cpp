// Scheduler.cpp -- task graph and worker threads of the gate scheduler (see Scheduler.h).
//
// Round 5: the state is PER DEVICE.  Every device of multiGPUs(n) has its own workers (CUHE_SCHED_THREADS per device, default
// 3), its own ready queues, staging groups and lock; a worker is bound to one device and owns one stream there.  The graph
// itself needs no global lock: a task counts its unissued dependencies in an atomic, the successor list and the "issued"
// flag of a task are guarded by the lock of the task's OWN device, reference counts are atomics, and what only the
// recording side touches (a node's last writer and readers) is guarded by the recording lock, which no worker ever takes.
// With 8 devices that is 24 host threads feeding 8 sets of hardware queues instead of three threads and one mutex
// ("the scheduler will not scale to the machine as written").
//
// Ordering on the GPU.  A task runs on the stream of the worker that picked it; a task it depends on may have run on another
// worker's stream (of this or of another device).  hipEventRecord is the expensive call here (5 us of host time with four
// threads launching, 18 us with eight: tools/ubench_launch.hip, profiles/r04_sched_prince.txt), so no task records an event of
// its own.  Every worker stream counts the tasks it has issued (`seq`) and owns ONE event; a consumer on another stream that
// needs "task #k of that stream has finished" records that event on the producer's stream only if its last record does not
// cover #k yet -- the record lands behind #k (and possibly behind later tasks: more ordering than asked for, never less) --
// and waits for it.  With the per-worker queues below most dependencies stay on one stream and need nothing at all.
//
// Batches.  A batchable gate whose dependencies have been issued waits in a staging group keyed by (kind, key); a worker
// that finds no regular task takes a group and runs it as ONE call of the batch runner.  Which group, and when, is the
// policy (CUHE_SCHED_POLICY): 0 = round 4 (the fullest group as soon as no regular task is in flight); 1 = groups that
// cannot grow any more first -- every recorded gate of that (kind, key) has arrived -- and an incomplete group only when
// nothing else on the device can make progress, the OLDEST one (the most upstream in the client's program order, so that
// the groups downstream fill up); 2 = as 1, but an incomplete group does not wait for workers that are inside batches.
// Under 1 and 2 a group whose gates are ready as the client records them (nothing is pending then: the first layer of a circuit)
// is taken only once the client has added nothing to it for CUHE_SCHED_QUIET_US (default 100); and a worker counts as busy
// until it has published the successors of what it ran.
#include "Scheduler.h"
#include "CuHE.h"
#include "Debug.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unistd.h>

namespace cuHE {
namespace sched { bool inWorker(); }
namespace detail {
// the reference's reaction to a failed device call is "message ; exit(-1)" (cuhe/Debug.h:35-66) from its one host thread.  From a worker
// thread of this library exit() would run the process's static destructors while the client's threads (and the other workers) still use
// what they destroy: the streams are flushed and the process ends with the same status at once.
void die(int code) {
	if (!sched::inWorker()) exit(code);
	fflush(NULL);
	_exit(code);
}
}
namespace sched {

// one worker's stream (on the worker's device)
struct StreamState {
	void *stream = nullptr, *event = nullptr;
	int dev = 0;
	std::atomic<long> seq{0};              // tasks issued on this stream so far (written by its worker only)
	std::mutex m;                          // guards `event` / `covered`
	long covered = 0;                      // the event's last record lies behind task #covered
	// what THIS stream has waited for already (its own worker only): per foreign stream, the `covered` of the event record it waited on last.
	// Everything enqueued here since is ordered behind that stream's tasks up to that number: no second wait (one barrier packet each) for them
	std::vector<std::pair<StreamState *, long>> waited;
};
struct Task {
	std::function<void(void *)> fn;
	int dev = 0;
	long id = 0;                           // recording order (all devices)
	std::vector<Task *> deps;              // every task this one is ordered after (references held until it has run); written before the task is published
	std::vector<Task *> succ;              // tasks that wait for this one to be issued        -- guarded by the lock of device `dev`
	std::vector<Node *> nodes;             // references held until it has run
	std::atomic<int> pending{1};           // dependencies not issued yet (+ 1 while submit() is still linking)
	std::atomic<bool> issued{false};       // set under the lock of device `dev`, after ss / seq
	StreamState *ss = nullptr; long seq = 0;     // where it ran: task #seq of that stream
	int kind = 0; long key = 0; Node *subject = nullptr, *op1 = nullptr, *op2 = nullptr;      // batchable gate (Scheduler.h)
	bool blocking = false;                 // ends with a wait on the host (a copy to or from the host and its synchronise): kept off the workers that take groups
	std::atomic<int> refs{1};              // the graph itself until the task has run; + nodes, successors, waiters
};

namespace {
typedef std::chrono::steady_clock clk;
constexpr int kMaxDevices = 64;

struct GroupKey { int kind; long key; bool operator<(const GroupKey &o) const { return kind != o.kind ? kind < o.kind : key < o.key; } };
struct Group { std::deque<Task *> q; clk::time_point lastArrival; bool fromClient = false; };     // lastArrival: of a gate that was ready when the client recorded it
// everything a device's workers share
struct DevState {
	int dev = 0;
	std::mutex m;                          // queues below + succ / issued of this device's tasks
	std::condition_variable cv;
	std::deque<Task *> ready;              // tasks that were ready when the client recorded them, or that another device's worker made ready
	std::vector<std::deque<Task *>> local; // per worker: tasks its own tasks made ready (newest at the back; thieves take the oldest)
	std::deque<Task *> blocking;           // ready tasks that block their worker on the host (uploads, x2z): for the workers that take no groups, when there are any
	std::map<GroupKey, Group> staged;      // batchable ready tasks by (kind, key)
	std::map<GroupKey, long> groupPending; // batchable tasks of that (kind, key) recorded but not staged yet
	long stagedCount = 0;
	int busyRegular = 0, busyBatch = 0;    // workers inside a regular task / inside a batch
	int batchIdle = 0;                     // workers that take groups and are waiting for work right now
	std::atomic<long> epoch{0};            // bumped whenever a waiting worker might find something new (work made ready, a busy count dropped): what a worker that
	                                       // SPINS for a moment before it sleeps watches -- a condition-variable wake-up costs 20-60 us, and a client that blocks once per
	                                       // S-box (the reference's Prince.cu: ZZX in, ZZX out) pays it on every step of the S-box's dependency chain
	std::vector<std::thread> workers;
	int started = 0;
	bool used = false;
	// statistics (under m)
	long batchesRun = 0, batchedTasks = 0, loneTasks = 0, waits = 0, records = 0, incompleteTaken = 0;
	double busySeconds = 0, idleSeconds = 0, gateSeconds = 0, orderSeconds = 0;
	std::map<int, std::map<int, long>> sizeHist;     // kind -> batch size -> count (CUHE_SCHED_TRACE)
};
DevState *devs[kMaxDevices];               // created on first use, never destroyed (tasks and idle streams point into them)
std::mutex devsMu;                         // creation of DevStates / workers
std::mutex recMu;                          // the recording side: node state (lastWrite, readers), task ids.  Never taken by a worker.
std::mutex doneMu;                         // waiters for "issued" / "outstanding == 0"
std::condition_variable cvDone;
std::atomic<long> outstanding{0}, totalTasks{0}, maxQueued{0}, nextId{0};
std::atomic<bool> active{false}, stopping{false};
extern std::atomic<long> issuedTotal;      // tasks issued so far (the watchdog's sign of life)
int workersPerDev = 3, stealing = 1, policy = 1, trace = 0;
int batchWorkers = 1;                      // how many of a device's workers take staged groups (0 = all; CUHE_SCHED_BATCH_WORKERS).  ONE: the batches of a device
                                           // follow each other on one stream with one set of batch scratch, the other workers run the regular tasks --
                                           // PRINCE gate by gate 0.074-0.085 s against 0.090-0.110 s with every worker taking groups (profiles/r05_sched_prince.txt)
long spinNs = 40000;                       // a worker with nothing to do polls its device's epoch for this long before it sleeps (CUHE_SCHED_SPIN_US; 0 = sleep at once)
std::atomic<int> clientsWaiting{0};        // client threads blocked in wait() / drain(): they record nothing while they wait, so a group they were feeding has stopped growing
std::atomic<int> resultWaiters{0};         // ... of them, those that wait for ONE result (x2z, a raw pointer: wait() / waitNode()), not for everything (drain()): while there
                                           // are any, the device runs for LATENCY -- a group goes as soon as a worker is free for it, whatever is still in flight (takeBatch)
long quietNs = 100000;                     // policies 1, 2: a group the CLIENT is adding ready gates to right now is taken only after it has been quiet for this long (CUHE_SCHED_QUIET_US)
BatchRunner batchRunner = nullptr; int maxBatch = 1;
thread_local bool tlsWorker = false;
thread_local void *tlsStream = nullptr;
thread_local StreamState *tlsSS = nullptr;         // this worker's stream (never freed: tasks point at it)
thread_local int tlsDev = -1, tlsIndex = -1;

void unrefTask(Task *t) { if (t->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete t; }
// objects whose node died are handed back to be deleted by the caller (their destructors call into the library)
void unrefNode(Node *n, std::vector<CuPolynomial *> &dead) {
	if (n->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
	if (n->lastWrite) unrefTask(n->lastWrite);           // nobody else holds the node any more
	for (Task *r : n->readers) unrefTask(r);
	if (n->obj) dead.push_back(n->obj);
	delete n;
}
// readers of a node that is never written (a key bit read by every round) would pile up: of the readers that have been
// issued, the last one per stream stands for the earlier ones of that stream
void pruneReaders(Node *n) {               // recMu held
	const std::vector<Task *> all = n->readers;
	std::vector<char> isIssued(all.size(), 0), shadowed(all.size(), 0);
	for (size_t i = 0; i < all.size(); ++i) isIssued[i] = all[i]->issued.load(std::memory_order_acquire) ? 1 : 0;    // (ss / seq are final once it reads true)
	for (size_t i = 0; i < all.size(); ++i) {
		if (!isIssued[i]) continue;
		Task *r = all[i];
		for (size_t j = 0; j < all.size() && !shadowed[i]; ++j) {
			Task *o = all[j];
			shadowed[i] = j != i && isIssued[j] && o->ss == r->ss && (o->seq > r->seq || (o->seq == r->seq && j > i));
		}
	}
	// one reference per ENTRY (submit() enters a task once per node, but the count must not depend on that: ADVICE r04)
	n->readers.clear();
	for (size_t i = 0; i < all.size(); ++i) if (!shadowed[i]) n->readers.push_back(all[i]);
	for (size_t i = 0; i < all.size(); ++i) if (shadowed[i]) unrefTask(all[i]);
}
// the stream of `me` (a worker's own, on device `dev`) waits until task #seq of `from` has finished, unless it has done so already;
// returns the number of events recorded (0 or 1), *waited: whether a wait was enqueued
std::atomic<long> waitsSkipped{0}, waitsDone{0};
int latencyOn = 1;                         // CUHE_SCHED_LATENCY=0: no latency mode, host-blocking tasks on any worker (A/B)
int waitDedup = 1, releaseOnlyOn = 1;      // CUHE_SCHED_WAIT_DEDUP / CUHE_SCHED_RELEASE_ONLY = 0: the behaviour before (A/B runs)
int orderAfter(int dev, StreamState *me, StreamState *from, long seq, bool *waited = nullptr) {
	std::pair<StreamState *, long> *mine = nullptr;
	for (auto &w : me->waited) if (w.first == from) { mine = &w; break; }
	if (waited) *waited = false;
	if (waitDedup && mine && mine->second >= seq) { waitsSkipped.fetch_add(1, std::memory_order_relaxed); return 0; }
	std::lock_guard<std::mutex> lk(from->m);
	int recorded = 0;
	if (from->covered < seq) {
		const long now = from->seq.load(std::memory_order_acquire);    // everything up to #now has been enqueued
		CSC(cuhe_hip_event_record(from->dev, from->event, from->stream));
		from->covered = now; recorded = 1;
	}
	CSC(cuhe_hip_stream_wait_event(dev, me->stream, from->event));
	if (mine) mine->second = from->covered; else me->waited.push_back({from, from->covered});
	if (waited) *waited = true;
	waitsDone.fetch_add(1, std::memory_order_relaxed);
	return recorded;
}

// ---- device blocks released and taken inside tasks.  The library's stream-ordered pool (cuhe_hip_malloc_stream) hands a
// block freed on one stream to another stream only behind everything enqueued on the first; here a released block carries
// "task #k of stream S" -- its last use -- and the taker is ordered behind exactly that, usually an event record of long ago.
// (a block released by a release-only task carries the positions of that task's dependencies: up to kMaxPos streams)
constexpr int kMaxPos = 3;
struct Pos { StreamState *ss; long seq; };
struct Block { void *ptr; Pos pos[kMaxPos]; int npos; };
// the release-only task this worker is running: the positions its blocks take (nullptr outside such a task) and, once a block that is
// NOT the cache's had to be released in the order of the worker's own stream, the fact that the stream has been ordered behind them
struct ReleaseCtx { int dev; StreamState *me; std::vector<std::pair<StreamState *, long>> *pos; bool ordered; long waits, records; };
thread_local ReleaseCtx *tlsRelease = nullptr;
struct BlockCache {                        // per device
	std::mutex m;
	std::unordered_map<size_t, std::deque<Block>> bySize;
	std::unordered_map<void *, size_t> sizeOf;           // blocks handed out by taskAlloc
	long hits = 0, foreign = 0, misses = 0;
	unsigned long long generation = 0;     // cuhe_hip_generation() the cached blocks belong to
	void checkGeneration() {               // m held: a cuhe_hip_shutdown since took every block with it
		const unsigned long long g = cuhe_hip_generation();
		if (g != generation) { bySize.clear(); sizeOf.clear(); generation = g; }
	}
};
BlockCache caches[kMaxDevices];
void flushCache() {                        // the devices have been synchronised: everything goes back to the library's pool
	for (int d = 0; d < kMaxDevices; ++d) {
		BlockCache &C = caches[d];
		std::lock_guard<std::mutex> lk(C.m);
		C.checkGeneration();
		for (auto &bs : C.bySize)
			for (Block &b : bs.second) { C.sizeOf.erase(b.ptr); CSC(cuhe_hip_free(d, b.ptr)); }
		C.bySize.clear();
	}
}

// D.m held: this worker's newest task, else the oldest the client recorded, else the oldest of the fullest other worker of the device
inline bool takesGroups(int me) { return batchWorkers <= 0 || me < batchWorkers; }
inline bool keepsOffBlocking(int me) { return batchWorkers > 0 && me < batchWorkers && workersPerDev > batchWorkers; }     // a group taker with colleagues that are not
Task *takeTask(DevState &D, int me) {
	if (!D.local[me].empty()) { Task *t = D.local[me].back(); D.local[me].pop_back(); return t; }
	if (!D.ready.empty()) { Task *t = D.ready.front(); D.ready.pop_front(); return t; }
	if (!D.blocking.empty() && !keepsOffBlocking(me)) { Task *t = D.blocking.front(); D.blocking.pop_front(); return t; }
	size_t best = 0; int from = -1;
	for (size_t w = 0; w < D.local.size(); ++w) if (D.local[w].size() > best) { best = D.local[w].size(); from = (int)w; }
	if (from < 0) return nullptr;
	Task *t = D.local[from].front(); D.local[from].pop_front();
	return t;
}
// D.m held: a staged group, up to maxBatch of its tasks (oldest first).  See the policies at the top of the file.
bool takeBatch(DevState &D, std::vector<Task *> &batch, long *retryNs) {
	*retryNs = 0;
	if (D.stagedCount == 0) return false;
	auto pick = D.staged.end();
	bool incomplete = false;
	if (policy == 0) {
		for (auto it = D.staged.begin(); it != D.staged.end(); ++it) if (pick == D.staged.end() || it->second.q.size() > pick->second.q.size()) pick = it;
		if (pick == D.staged.end() || pick->second.q.empty()) return false;
		if ((int)pick->second.q.size() < maxBatch && D.busyRegular > 0) return false;
	} else {
		// a full group goes at once
		for (auto it = D.staged.begin(); it != D.staged.end(); ++it) if ((int)it->second.q.size() >= maxBatch) { pick = it; break; }
		if (pick == D.staged.end()) {
			// complete groups (nothing recorded for them is still on its way, so they cannot grow): the fullest
			for (auto it = D.staged.begin(); it != D.staged.end(); ++it) {
				if (it->second.q.empty()) continue;
				auto gp = D.groupPending.find(it->first);
				if (gp != D.groupPending.end() && gp->second > 0) continue;
				if (pick == D.staged.end() || it->second.q.size() > pick->second.q.size()) pick = it;
			}
		}
		if (pick == D.staged.end()) {
			// only incomplete groups.  What a regular task in flight makes ready may belong to any of them; so may what the
			// batch in flight on another worker makes ready (policy 1 waits for that too)
			// -- unless a client is blocked on a result: then the oldest group goes now (the worker asking is free, and what is in flight may be the
			// upload or the copy down of ANOTHER client thread, hundreds of microseconds on the host; groups still fill up while the takers are busy)
			const bool latency = latencyOn && resultWaiters.load(std::memory_order_relaxed) > 0;
			if (D.busyRegular > 0 && !latency) return false;
			if (policy == 1 && D.busyBatch > 0 && !latency) return false;
			for (auto it = D.staged.begin(); it != D.staged.end(); ++it) {
				if (it->second.q.empty()) continue;
				if (pick == D.staged.end() || it->second.q.front()->id < pick->second.q.front()->id) pick = it;
			}
			incomplete = true;
		}
		if (pick == D.staged.end()) return false;
		// the client is recording this very group (its gates are ready as they are recorded, so nothing is "pending" -- the first layers
		// of a circuit, before the client is ahead of the workers): a group that is not full waits until it has been quiet for quietNs
		if ((int)pick->second.q.size() < maxBatch && quietNs > 0 && pick->second.fromClient && clientsWaiting.load(std::memory_order_relaxed) == 0) {
			const long age = (long)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - pick->second.lastArrival).count();
			if (age < quietNs) { *retryNs = quietNs - age; return false; }
		}
	}
	std::deque<Task *> &q = pick->second.q;
	while (!q.empty() && (int)batch.size() < maxBatch) { batch.push_back(q.front()); q.pop_front(); --D.stagedCount; }
	if (q.empty()) D.staged.erase(pick);
	if (incomplete) ++D.incompleteTaken;
	return true;
}
DevState &deviceState(int dev);
// a task whose dependencies have all been issued.  `me` / `next`: the calling worker's index on ITS device and its chain slot
void makeReady(Task *t, Task **next) {
	DevState &D = deviceState(t->dev);
	if (t->kind > 0 && batchRunner && maxBatch > 1) {
		std::lock_guard<std::mutex> lk(D.m);
		const GroupKey k{t->kind, t->key};
		Group &G = D.staged[k];
		G.q.push_back(t); ++D.stagedCount;
		if (!tlsWorker) { G.lastArrival = clk::now(); G.fromClient = true; }
		auto gp = D.groupPending.find(k);
		if (gp != D.groupPending.end() && --gp->second <= 0) D.groupPending.erase(gp);
		// the one woken must be a worker that takes groups: with a restricted set, everybody is woken -- but only when one of that set
		// is waiting at all (it is busy most of the time and finds the group when it comes back; a broadcast per staged gate showed as
		// 5 % of the busy worker's time in the sampling profile)
		D.epoch.fetch_add(1, std::memory_order_release);
		if (batchWorkers <= 0) D.cv.notify_one(); else if (D.batchIdle > 0) D.cv.notify_all();
		return;
	}
	if (t->blocking && latencyOn && workersPerDev > batchWorkers && batchWorkers > 0) {        // to a worker that takes no groups (whoever is woken may be one that does: wake all)
		std::lock_guard<std::mutex> lk(D.m);
		D.blocking.push_back(t);
		D.epoch.fetch_add(1, std::memory_order_release);
		D.cv.notify_all();
		return;
	}
	if (next && !*next && t->dev == tlsDev) { *next = t; return; }      // follow the chain on this stream: no event wait, warm scratch
	std::lock_guard<std::mutex> lk(D.m);
	if (tlsWorker && t->dev == tlsDev && stealing && tlsIndex >= 0 && tlsIndex < (int)D.local.size()) D.local[tlsIndex].push_back(t);
	else D.ready.push_back(t);
	D.epoch.fetch_add(1, std::memory_order_release);
	D.cv.notify_one();
}
// streams of workers that have gone (setScheduled(false) ; setScheduled(true) cycles): a stream costs ~10 ms to create, and issued
// tasks keep pointing at their StreamState, so the states are never destroyed -- the next workers take them over
std::mutex idleMu;
std::vector<StreamState *> idleStreams[kMaxDevices];
// a worker launches every kernel of its device: it runs on the CPUs local to that device (include/cuhe_hip.h, cuhe_hip_pin_thread_to_device: blocks
// issued from the far socket of a two-socket host take 0.065-0.070 s against 0.058-0.059 s); CUHE_SCHED_PIN=0 leaves the placement to the kernel
int pinWorkers = 1;
StreamState *acquireStream(int dev) {
	if (pinWorkers && tlsWorker) cuhe_hip_pin_thread_to_device(dev);
	{
		std::lock_guard<std::mutex> lk(idleMu);
		if (!idleStreams[dev].empty()) { StreamState *s = idleStreams[dev].back(); idleStreams[dev].pop_back(); return s; }
	}
	StreamState *ns = new StreamState;
	ns->dev = dev;
	CSC(cuhe_hip_stream_create(dev, &ns->stream));
	CSC(cuhe_hip_event_create(dev, &ns->event));
	return ns;
}
struct ReturnStream { ~ReturnStream() {                   // at worker exit
	if (!tlsSS) return;
	std::lock_guard<std::mutex> lk(idleMu);
	idleStreams[tlsSS->dev].push_back(tlsSS);
	tlsSS = nullptr;
} };

void workerMain(DevState *Dp, int me) {
	DevState &D = *Dp;
	tlsWorker = true; tlsDev = D.dev; tlsIndex = me;
	ReturnStream giveBack;
	// a stream costs ~10 ms to create: before the first task, not inside it (when the library is up already)
	if (cuhe_hip_is_initialised() && D.dev < cuhe_hip_num_gpus()) tlsSS = acquireStream(D.dev);
	std::unique_lock<std::mutex> lk(D.m);
	++D.started; D.cv.notify_all();
	{ std::lock_guard<std::mutex> dl(doneMu); }
	cvDone.notify_all();
	Task *next = nullptr;
	std::vector<Task *> batch, succAll;
	for (;;) {
		batch.clear();
		if (next) { batch.push_back(next); next = nullptr; }
		else {
			const auto w0 = clk::now();
			for (;;) {
				if (Task *t = takeTask(D, me)) { batch.push_back(t); break; }
				long retryNs = 0;
				if (takesGroups(me) && takeBatch(D, batch, &retryNs)) break;
				if (stopping.load()) return;
				if (spinNs > 0) {                  // poll for a moment (lock released) before going to sleep
					const long seen = D.epoch.load(std::memory_order_acquire);
					lk.unlock();
					const auto until = clk::now() + std::chrono::nanoseconds(retryNs > 0 && retryNs < spinNs ? retryNs : spinNs);
					bool changed = false;
					while (!(changed = D.epoch.load(std::memory_order_acquire) != seen) && !stopping.load(std::memory_order_relaxed) && clk::now() < until) __builtin_ia32_pause();
					lk.lock();
					// (a bump or a stop between the last poll and the lock would otherwise be slept through: every bump happens before its notify)
					if (stopping.load()) return;
					if (changed || D.epoch.load(std::memory_order_acquire) != seen || (retryNs > 0 && retryNs <= spinNs)) continue;
				}
				const bool groupTaker = takesGroups(me);
				if (groupTaker) ++D.batchIdle;
				if (retryNs > 0) D.cv.wait_for(lk, std::chrono::nanoseconds(retryNs)); else D.cv.wait(lk);
				if (groupTaker) --D.batchIdle;
			}
			D.idleSeconds += std::chrono::duration<double>(clk::now() - w0).count();
		}
		const auto b0 = clk::now();
		const bool asBatch = batch[0]->kind > 0 && batchRunner && maxBatch > 1;       // (a group of one member still counts as a batch in flight)
		if (asBatch) ++D.busyBatch; else ++D.busyRegular;
		D.used = true;
		lk.unlock();
		if (!tlsSS) tlsSS = acquireStream(D.dev);
		StreamState *ss = tlsSS;
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
		const bool releaseOnly = releaseOnlyOn && batch.size() == 1 && batch[0]->kind == kReleaseOnly;
		ReleaseCtx rel{D.dev, ss, &latest, false, 0, 0};
		if (releaseOnly) {                      // no wait here: the blocks it releases take the positions of its dependencies (own stream included)
			for (Task *d : batch[0]->deps) {
				if (d->ss != ss) continue;
				bool found = false;
				for (auto &e : latest) if (e.first == ss) { if (d->seq > e.second) e.second = d->seq; found = true; }
				if (!found) latest.push_back({ss, d->seq});
			}
			tlsRelease = &rel;
		} else
			for (auto &e : latest) { bool w; records += orderAfter(D.dev, ss, e.first, e.second, &w); if (w) ++waits; }
		tlsStream = s;
		const auto f0 = clk::now();
		if (releaseOnly) { batch[0]->fn(s); tlsRelease = nullptr; waits += rel.waits; records += rel.records; }
		else if (batch.size() == 1) batch[0]->fn(s);
		else {
			std::vector<Node *> subjects, o1, o2;
			for (Task *t : batch) { subjects.push_back(t->subject); o1.push_back(t->op1); o2.push_back(t->op2); }
			batchRunner(batch[0]->kind, subjects.data(), o1.data(), o2.data(), (int)subjects.size(), s);
		}
		for (Task *t : batch) t->fn = nullptr;      // the closures' captures go before any lock is taken again
		const auto f1 = clk::now();
		const long seq = ss->seq.load(std::memory_order_relaxed) + 1;
		ss->seq.store(seq, std::memory_order_release);
		// publish: successors recorded from now on find the tasks issued; those recorded before are collected
		succAll.clear();
		lk.lock();
		for (Task *t : batch) {
			t->ss = ss; t->seq = seq;
			t->issued.store(true, std::memory_order_release);
			succAll.insert(succAll.end(), t->succ.begin(), t->succ.end());
			t->succ.clear();
		}
		if (batch.size() > 1) { ++D.batchesRun; D.batchedTasks += (long)batch.size(); } else ++D.loneTasks;
		if (trace && batch[0]->kind > 0) ++D.sizeHist[batch[0]->kind][(int)batch.size()];
		D.waits += waits; D.records += records;
		D.busySeconds += std::chrono::duration<double>(clk::now() - b0).count();
		D.gateSeconds += std::chrono::duration<double>(f1 - f0).count();
		D.orderSeconds += std::chrono::duration<double>(f0 - b0).count();
		lk.unlock();
		std::vector<CuPolynomial *> dead;
		for (Task *t : batch) {
			for (Task *d : t->deps) unrefTask(d);
			t->deps.clear();
			for (Node *n : t->nodes) unrefNode(n, dead);
			t->nodes.clear();
		}
		for (Task *x : succAll) if (x->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) makeReady(x, &next);
		outstanding.fetch_sub((long)batch.size(), std::memory_order_acq_rel);
		issuedTotal.fetch_add((long)batch.size(), std::memory_order_relaxed);
		{ std::lock_guard<std::mutex> dl(doneMu); }
		cvDone.notify_all();
		for (Task *t : batch) unrefTask(t);
		for (CuPolynomial *p : dead) delete p;
		lk.lock();
		if (asBatch) --D.busyBatch; else --D.busyRegular;                // (only now: what this task made ready has been staged)
		D.epoch.fetch_add(1, std::memory_order_release);
		if (D.stagedCount && D.batchIdle > 0 && (D.busyRegular == 0 || batchWorkers > 0)) D.cv.notify_all();     // groups that waited for the work in flight to drain
	}
}

DevState &deviceState(int dev) {
	if (dev < 0) dev = 0;
	if (dev >= kMaxDevices) { fprintf(stderr, "scheduler: device %d out of range\n", dev); exit(-1); }
	DevState *D = devs[dev];
	if (D) return *D;
	std::lock_guard<std::mutex> lk(devsMu);
	if (!devs[dev]) { DevState *n = new DevState; n->dev = dev; devs[dev] = n; }
	return *devs[dev];
}
// workers of a device: started when the mode is switched on (devices known then) or at the first task recorded for the device
void ensureWorkers(int dev) {
	DevState &D = deviceState(dev);
	{
		std::lock_guard<std::mutex> lk(D.m);
		if (!D.workers.empty()) return;
	}
	std::lock_guard<std::mutex> cl(devsMu);
	std::unique_lock<std::mutex> lk(D.m);
	if (!D.workers.empty()) return;
	D.local.assign(workersPerDev, std::deque<Task *>());
	D.started = 0;
	for (int i = 0; i < workersPerDev; ++i) D.workers.emplace_back(workerMain, &D, i);
	const int n = workersPerDev;
	D.cv.wait(lk, [&D, n] { return D.started == n; });      // their streams exist
}

// ---- watchdog: a wait that sees no task issued for CUHE_SCHED_WATCHDOG_S seconds (default 60; 0 = never) reports the
// queues of every device to stderr -- once per period -- instead of hanging silently; the wait itself goes on.
std::atomic<long> issuedTotal{0};
long watchdogSeconds() { static const long s = getenv("CUHE_SCHED_WATCHDOG_S") ? atol(getenv("CUHE_SCHED_WATCHDOG_S")) : 60; return s; }
void dumpState(const char *where, Task *stuck) {
	fprintf(stderr, "scheduler watchdog: %s has seen no task issued for %ld s; %ld task(s) outstanding, %ld recorded, %ld issued, policy %d, %d worker(s) per device, %d take groups\n",
	        where, watchdogSeconds(), outstanding.load(), totalTasks.load(), issuedTotal.load(), policy, workersPerDev, batchWorkers);
	if (stuck) {
		fprintf(stderr, "  waiting for task #%ld (kind %d key %ld device %d): issued %d, %d dependencies not issued, depends on:", stuck->id, stuck->kind, stuck->key, stuck->dev,
		        (int)stuck->issued.load(), stuck->pending.load());
		for (Task *d : stuck->deps) fprintf(stderr, " #%ld(kind %d, %s)", d->id, d->kind, d->issued.load() ? "issued" : "NOT issued");
		fprintf(stderr, "\n");
	}
	for (int d = 0; d < kMaxDevices; ++d) {
		DevState *D = devs[d];
		if (!D) continue;
		std::unique_lock<std::mutex> lk(D->m, std::try_to_lock);
		if (!lk.owns_lock()) { fprintf(stderr, "  device %d: its lock is HELD (a worker is inside the queues)\n", d); continue; }
		if (D->workers.empty() && !D->used) continue;
		size_t local = 0; for (auto &q : D->local) local += q.size();
		fprintf(stderr, "  device %d: %zu worker(s), ready %zu, host-blocking %zu, local %zu, staged %ld in %zu group(s), busy regular %d / batch %d, group takers idle %d\n", d, D->workers.size(),
		        D->ready.size(), D->blocking.size(), local, D->stagedCount, D->staged.size(), D->busyRegular, D->busyBatch, D->batchIdle);
		for (auto &g : D->staged) {
			auto gp = D->groupPending.find(g.first);
			fprintf(stderr, "    group kind %d key %ld: %zu staged (oldest #%ld), %ld still on their way%s\n", g.first.kind, g.first.key, g.second.q.size(),
			        g.second.q.empty() ? -1L : g.second.q.front()->id, gp == D->groupPending.end() ? 0L : gp->second, g.second.fromClient ? ", fed by the client" : "");
		}
		for (auto &gp : D->groupPending) if (!D->staged.count(gp.first)) fprintf(stderr, "    group kind %d key %ld: nothing staged, %ld on their way\n", gp.first.kind, gp.first.key, gp.second);
	}
	fflush(stderr);
}
// cvDone.wait with the watchdog; doneMu held through lk
struct ClientWaiting {                     // a client thread inside wait() / drain(): groups it was feeding have stopped growing -- no "quiet" delay for them
	const bool result;                     // ... inside wait(): for one result (resultWaiters)
	explicit ClientWaiting(bool oneResult) : result(oneResult) {
		const bool first = clientsWaiting.fetch_add(1, std::memory_order_acq_rel) == 0;
		const bool firstResult = result && resultWaiters.fetch_add(1, std::memory_order_acq_rel) == 0;
		if (first || firstResult)            // what the group takers decided not to take may go now
			for (DevState *D : devs) if (D) { D->epoch.fetch_add(1, std::memory_order_release); { std::lock_guard<std::mutex> lk(D->m); } D->cv.notify_all(); }
	}
	~ClientWaiting() { if (result) resultWaiters.fetch_sub(1, std::memory_order_acq_rel); clientsWaiting.fetch_sub(1, std::memory_order_acq_rel); }
};
template <typename Pred> void waitDone(std::unique_lock<std::mutex> &lk, const char *where, Task *stuck, Pred pred) {
	if (pred()) return;
	if (tlsWorker) { cvDone.wait(lk, pred); return; }          // (never: workers do not wait on tasks)
	lk.unlock();
	ClientWaiting here(stuck != nullptr);
	if (spinNs > 0) {                          // the answer is usually microseconds away: poll before sleeping
		const auto until = clk::now() + std::chrono::nanoseconds(4 * spinNs);
		while (!pred() && clk::now() < until) __builtin_ia32_pause();
	}
	lk.lock();
	if (pred()) return;
	const long limit = watchdogSeconds();
	if (limit <= 0) { cvDone.wait(lk, pred); return; }
	long seen = issuedTotal.load(std::memory_order_relaxed);
	auto since = clk::now();
	while (!pred()) {
		cvDone.wait_for(lk, std::chrono::seconds(1));
		const long now = issuedTotal.load(std::memory_order_relaxed);
		if (now != seen) { seen = now; since = clk::now(); continue; }
		if (std::chrono::duration_cast<std::chrono::seconds>(clk::now() - since).count() >= limit && !pred()) {
			lk.unlock(); dumpState(where, stuck); lk.lock();
			since = clk::now();
		}
	}
}

struct AtExit { ~AtExit() {                 // idle workers must not outlive the process's static state
	if (tlsWorker) { for (DevState *D : devs) if (D) for (auto &w : D->workers) w.detach(); return; }      // exit(-1) from a failed call inside a task: nothing to wait for
	stopping.store(true);
	for (DevState *D : devs) if (D) { { std::lock_guard<std::mutex> lk(D->m); } D->cv.notify_all(); }
	for (DevState *D : devs) if (D) { for (auto &w : D->workers) if (w.joinable()) w.join(); D->workers.clear(); }
} } atExit;
} // namespace

bool on() { return active.load(std::memory_order_acquire); }
bool inWorker() { return tlsWorker; }
void *workerStream() { return tlsStream; }
int threads() { int n = 0; for (DevState *D : devs) if (D) n += (int)D->workers.size(); return n; }

void start(int n) {
	if (active.load()) return;
	std::lock_guard<std::mutex> rl(recMu);
	if (active.load()) return;
	if (n <= 0) { const char *e = getenv("CUHE_SCHED_THREADS"); n = e ? atoi(e) : 0; }
	if (n <= 0) n = 3;                          // per device.  PRINCE gate by gate on one device: profiles/r05_sched_prince.txt
	workersPerDev = n;
	stopping.store(false);
	if (getenv("CUHE_SCHED_LOCAL")) stealing = atoi(getenv("CUHE_SCHED_LOCAL"));
	if (getenv("CUHE_SCHED_POLICY")) policy = atoi(getenv("CUHE_SCHED_POLICY"));
	if (getenv("CUHE_SCHED_QUIET_US")) quietNs = 1000L * atol(getenv("CUHE_SCHED_QUIET_US"));
	if (getenv("CUHE_SCHED_SPIN_US")) spinNs = 1000L * atol(getenv("CUHE_SCHED_SPIN_US"));
	if (getenv("CUHE_SCHED_BATCH_WORKERS")) batchWorkers = atoi(getenv("CUHE_SCHED_BATCH_WORKERS"));
	if (getenv("CUHE_SCHED_WAIT_DEDUP")) waitDedup = atoi(getenv("CUHE_SCHED_WAIT_DEDUP"));
	if (getenv("CUHE_SCHED_RELEASE_ONLY")) releaseOnlyOn = atoi(getenv("CUHE_SCHED_RELEASE_ONLY"));
	if (getenv("CUHE_SCHED_LATENCY")) latencyOn = atoi(getenv("CUHE_SCHED_LATENCY"));
	if (getenv("CUHE_SCHED_PIN")) pinWorkers = atoi(getenv("CUHE_SCHED_PIN"));
	trace = getenv("CUHE_SCHED_TRACE") ? atoi(getenv("CUHE_SCHED_TRACE")) : 0;
	const int nd = std::max(1, std::min(cuhe_hip_num_gpus(), kMaxDevices));
	for (int d = 0; d < nd; ++d) ensureWorkers(d);
	active.store(true, std::memory_order_release);
}
void drain() {
	{
		std::unique_lock<std::mutex> lk(doneMu);
		waitDone(lk, "drain()", nullptr, [] { return outstanding.load(std::memory_order_acquire) == 0; });
	}
	for (int d = 0; d < kMaxDevices; ++d) {
		DevState *D = devs[d];
		if (!D) continue;
		bool used;
		{ std::lock_guard<std::mutex> lk(D->m); used = D->used; }
		if (used && cuhe_hip_is_initialised() && d < cuhe_hip_num_gpus()) CSC(cuhe_hip_device_sync(d));
	}
	flushCache();
}
void *taskAlloc(int dev, size_t bytes) {
	StreamState *me = tlsSS && tlsSS->dev == dev ? tlsSS : nullptr;
	if (dev < 0 || dev >= kMaxDevices) return cuhe_hip_malloc(dev, bytes);
	BlockCache &C = caches[dev];
	Block b{nullptr, {}, 0};
	//  a thread that has no stream on `dev` -- work on another device inside a task (after moveTo / copyTo), a thread that is not a
	// worker -- cannot be ordered behind a cached block's last use: it takes a fresh block from the library, never one of the cache's
	if (me) {
		std::lock_guard<std::mutex> lk(C.m);
		C.checkGeneration();
		auto it = C.bySize.find(bytes);
		if (it != C.bySize.end() && !it->second.empty()) {
			std::deque<Block> &q = it->second;
			size_t pick = q.size();
			auto own = [&](const Block &x) { for (int k = 0; k < x.npos; ++k) if (x.pos[k].ss != me) return false; return true; };
			for (size_t i = q.size(); i-- > 0 && q.size() - i <= 8;) if (own(q[i])) { pick = i; break; }     // last used on this stream only: nothing to wait for
			if (pick == q.size()) { pick = 0; ++C.foreign; } else ++C.hits;                                // else the one released longest ago
			b = q[pick]; q.erase(q.begin() + pick);
		} else ++C.misses;
	}
	if (b.ptr) {
		for (int k = 0; k < b.npos; ++k) if (b.pos[k].ss != me) orderAfter(dev, me, b.pos[k].ss, b.pos[k].seq);
		return b.ptr;
	}
	void *p = cuhe_hip_malloc(dev, bytes);
	if (!p) return nullptr;
	std::lock_guard<std::mutex> lk(C.m);
	C.sizeOf[p] = bytes;
	return p;
}
// a block the CLIENT thread took from the library for a polynomial that now enters the graph (the upload of a host value, CuHE.cpp: hostValueUp):
// nothing is pending on it, and the task that releases it may keep it for the next taker like one of taskAlloc's -- otherwise it goes back through
// the library's stream-ordered pool and the next stream to take it waits for the whole stream that released it (39 such hand-overs per PRINCE block)
void adoptBlock(int dev, void *p, size_t bytes) {
	if (!p || dev < 0 || dev >= kMaxDevices || !active.load(std::memory_order_acquire)) return;
	BlockCache &C = caches[dev];
	std::lock_guard<std::mutex> lk(C.m);
	C.checkGeneration();
	C.sizeOf[p] = bytes;
}
void forgetBlock(void *p) {                // released outside a task (the client thread, after a detach): the library owns it again
	for (int d = 0; d < kMaxDevices; ++d) {
		if (!devs[d]) continue;               // (blocks are only ever handed out on devices that have run tasks)
		BlockCache &C = caches[d];
		std::lock_guard<std::mutex> lk(C.m);
		if (!C.sizeOf.empty()) C.sizeOf.erase(p);
	}
}
// a release-only task has to put its worker's stream behind its dependencies after all (a block that is not the cache's goes back in
// the order of that stream; more streams than a block has room for)
static void orderReleaseStream(ReleaseCtx *R) {
	if (R->ordered) return;
	for (auto &e : *R->pos) if (e.first != R->me) { bool w; R->records += orderAfter(R->dev, R->me, e.first, e.second, &w); if (w) ++R->waits; }
	R->ordered = true;
}
bool taskFree(int dev, void *p) {
	StreamState *me = tlsSS && tlsSS->dev == dev ? tlsSS : nullptr;
	if (dev < 0 || dev >= kMaxDevices) return false;
	ReleaseCtx *R = tlsRelease;
	BlockCache &C = caches[dev];
	std::unique_lock<std::mutex> lk(C.m);
	C.checkGeneration();
	auto it = C.sizeOf.find(p);
	if (it == C.sizeOf.end() || !me) {                              // not one of ours (allocated before the object was attached), or no stream here to
		if (it != C.sizeOf.end()) C.sizeOf.erase(it);                 // date its last use with: the library takes it back, and must not find a stale size
		lk.unlock();                                                  // here when it hands the pointer out again
		if (R) orderReleaseStream(R);
		return false;
	}
	const size_t bytes = it->second;
	if (R && !R->ordered && R->dev == dev && !R->pos->empty() && (int)R->pos->size() <= kMaxPos) {
		Block b{p, {}, 0};                                            // last uses: the dependencies of the running task, wherever they ran
		for (auto &e : *R->pos) b.pos[b.npos++] = Pos{e.first, e.second};
		C.bySize[bytes].push_back(b);
		return true;
	}
	if (R && !R->ordered) { lk.unlock(); orderReleaseStream(R); lk.lock(); }
	Block b{p, {}, 1};
	b.pos[0] = Pos{me, me->seq.load(std::memory_order_relaxed) + 1};  // last use: the running task
	C.bySize[bytes].push_back(b);
	return true;
}
void stop() {
	if (!active.load()) return;
	drain();
	active.store(false, std::memory_order_release);
	stopping.store(true);
	for (DevState *D : devs) if (D) { { std::lock_guard<std::mutex> lk(D->m); } D->cv.notify_all(); }
	int nworkers = 0, ndev = 0;
	for (DevState *D : devs) if (D) {
		for (auto &w : D->workers) w.join();       // (their streams stay with the library: blocks parked on them settle as they go idle)
		nworkers += (int)D->workers.size(); if (!D->workers.empty()) ++ndev;
		D->workers.clear();
	}
	if (getenv("CUHE_SCHED_STATS") && atoi(getenv("CUHE_SCHED_STATS")) != 0) {
		long long ac[4] = {0, 0, 0, 0};
		cuhe_hip_alloc_counters(ac);
		printf("allocator: %lld hipMalloc, %lld pool hits, %lld stream hits, %lld cross-stream hand-overs\n", ac[0], ac[1], ac[2], ac[3]);
		long hits = 0, foreign = 0, misses = 0;
		for (BlockCache &C : caches) { hits += C.hits; foreign += C.foreign; misses += C.misses; }
		printf("task blocks: %ld from the same stream, %ld from another stream (ordered behind their last use), %ld from the library; stream waits enqueued %ld, "
		       "not needed (already behind that task) %ld\n", hits, foreign, misses, waitsDone.load(), waitsSkipped.load());
		long batches = 0, batched = 0, lone = 0, waits = 0, records = 0, incomplete = 0;
		double busy = 0, gate = 0, order = 0, idle = 0;
		for (DevState *D : devs) if (D) {
			batches += D->batchesRun; batched += D->batchedTasks; lone += D->loneTasks; waits += D->waits; records += D->records; incomplete += D->incompleteTaken;
			busy += D->busySeconds; gate += D->gateSeconds; order += D->orderSeconds; idle += D->idleSeconds;
		}
		printf("batches: %ld calls of the batch runner for %ld gates (%.1f per call; %ld groups taken before they were complete); %ld tasks ran alone; policy %d\n",
		       batches, batched, batches ? (double)batched / batches : 0.0, incomplete, lone, policy);
		printf("scheduler: %ld tasks, %ld cross-stream waits on %ld event records, at most %ld tasks recorded ahead; %d workers on %d device(s) busy %.3f s (%.3f in the gates, %.3f ordering streams), idle %.3f s in total\n",
		       totalTasks.load(), waits, records, maxQueued.load(), nworkers, ndev, busy, gate, order, idle);
		if (trace) {
			static const char *names[] = {"-", "x2c", "x2n", "relin", "modSwitch", "cAnd", "cXor", "copy", "cNot"};
			for (DevState *D : devs) if (D) for (auto &kh : D->sizeHist) {
				long calls = 0, members = 0;
				for (auto &sc : kh.second) { calls += sc.second; members += (long)sc.first * sc.second; }
				printf("  device %d %-9s %5ld calls %6ld gates:", D->dev, kh.first >= 0 && kh.first <= 8 ? names[kh.first] : "?", calls, members);
				for (auto &sc : kh.second) printf(" %dx%ld", sc.first, sc.second);
				printf("\n");
			}
		}
	}
	for (DevState *D : devs) if (D) {
		D->batchesRun = D->batchedTasks = D->loneTasks = D->waits = D->records = D->incompleteTaken = 0;
		D->busySeconds = D->idleSeconds = D->gateSeconds = D->orderSeconds = 0; D->sizeHist.clear(); D->used = false;
	}
	for (BlockCache &C : caches) C.hits = C.foreign = C.misses = 0;
	waitsDone = 0; waitsSkipped = 0;
	totalTasks.store(0); maxQueued.store(0);
}

Node *newNode(CuPolynomial *obj) { Node *n = new Node; n->obj = obj; return n; }
void releaseNode(Node *n) {
	std::vector<CuPolynomial *> dead;
	unrefNode(n, dead);
	for (CuPolynomial *p : dead) delete p;
}

void setBatchRunner(BatchRunner r, int mb) {
	std::lock_guard<std::mutex> lk(recMu);
	const char *e = getenv("CUHE_SCHED_BATCH");
	batchRunner = (e && atoi(e) == 0) ? nullptr : r;
	maxBatch = e && atoi(e) > 1 ? atoi(e) : mb;
	if (maxBatch > mb) maxBatch = mb;
}
Task *submit(int dev, const std::vector<Node *> &reads, const std::vector<Node *> &writes, std::function<void(void *)> fn, bool keep, int kind, long key, Node *subject,
             Node *op1, Node *op2) {
	if (dev < 0) dev = 0;                       // (an object that was never placed: its task only releases host state)
	ensureWorkers(dev);
	Task *t = new Task;
	if (kind == kHostBlocking) { t->blocking = true; kind = 0; }
	if (keep && kind == 0) t->blocking = true;  // (the client waits for it: x2z ends with the copy down and its synchronise; a batchable gate somebody waits for is still a gate)
	t->fn = std::move(fn); t->dev = dev; t->kind = kind; t->key = key; t->subject = subject; t->op1 = op1; t->op2 = op2;
	const bool batchable = kind > 0 && batchRunner && maxBatch > 1;
	{
		std::lock_guard<std::mutex> lk(recMu);
		t->id = nextId.fetch_add(1) + 1;
		auto after = [&](Task *d) {
			if (!d) return;
			for (Task *e : t->deps) if (e == d) return;
			d->refs.fetch_add(1, std::memory_order_relaxed); t->deps.push_back(d);
			DevState &P = deviceState(d->dev);
			std::lock_guard<std::mutex> dl(P.m);
			if (!d->issued.load(std::memory_order_acquire)) { d->succ.push_back(t); t->pending.fetch_add(1, std::memory_order_relaxed); }
		};
		auto written = [&](Node *n) { return std::find(writes.begin(), writes.end(), n) != writes.end(); };
		auto held = [&](Node *n) { return std::find(t->nodes.begin(), t->nodes.end(), n) != t->nodes.end(); };
		for (Node *r : reads) after(r->lastWrite);
		for (Node *w : writes) { after(w->lastWrite); for (Task *r : w->readers) after(r); }
		for (Node *r : reads) {
			if (written(r)) continue;
			if (!r->readers.empty() && r->readers.back() == t) continue;      // listed twice (cAnd(out, x, x)): one entry, one reference
			if (r->readers.size() >= 24) pruneReaders(r);
			r->readers.push_back(t); t->refs.fetch_add(1, std::memory_order_relaxed);
		}
		for (Node *w : writes) {
			if (held(w)) continue;                  // (listed twice)
			for (Task *r : w->readers) unrefTask(r);
			w->readers.clear();
			if (w->lastWrite) unrefTask(w->lastWrite);
			w->lastWrite = t; t->refs.fetch_add(1, std::memory_order_relaxed);
			w->refs.fetch_add(1, std::memory_order_relaxed); t->nodes.push_back(w);
		}
		for (Node *r : reads) if (!held(r)) { r->refs.fetch_add(1, std::memory_order_relaxed); t->nodes.push_back(r); }
		if (keep) t->refs.fetch_add(1, std::memory_order_relaxed);
	}
	const long out = outstanding.fetch_add(1, std::memory_order_acq_rel) + 1;
	totalTasks.fetch_add(1, std::memory_order_relaxed);
	long mq = maxQueued.load(std::memory_order_relaxed);
	while (out > mq && !maxQueued.compare_exchange_weak(mq, out)) {}
	if (batchable) {                            // one more gate of this (kind, key) is on its way to the device's staging area
		DevState &D = deviceState(dev);
		std::lock_guard<std::mutex> dl(D.m);
		++D.groupPending[GroupKey{kind, key}];
	}
	if (t->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) makeReady(t, nullptr);       // (the linking guard goes: the task is published)
	return t;
}

// the client needs the task's device work finished: an event of its own behind the task on the task's stream
void wait(Task *t) {
	{
		std::unique_lock<std::mutex> lk(doneMu);
		waitDone(lk, "wait()", t, [t] { return t->issued.load(std::memory_order_acquire); });
	}
	StreamState *ss = t->ss;
	thread_local std::vector<void *> mine;      // per client thread and device
	if ((int)mine.size() <= ss->dev) mine.resize(ss->dev + 1, nullptr);
	if (!mine[ss->dev]) CSC(cuhe_hip_event_create(ss->dev, &mine[ss->dev]));
	CSC(cuhe_hip_event_record(ss->dev, mine[ss->dev], ss->stream));
	CSC(cuhe_hip_event_sync(ss->dev, mine[ss->dev]));
	unrefTask(t);
}
void waitNode(Node *n) {
	std::vector<Task *> ts;
	{
		std::lock_guard<std::mutex> lk(recMu);
		if (n->lastWrite) { n->lastWrite->refs.fetch_add(1, std::memory_order_relaxed); ts.push_back(n->lastWrite); }
		for (Task *r : n->readers) { r->refs.fetch_add(1, std::memory_order_relaxed); ts.push_back(r); }
	}
	for (Task *t : ts) wait(t);
}
Stats stats() { return Stats{totalTasks.load(), 0, maxQueued.load()}; }

} // namespace sched
} // namespace cuHE
