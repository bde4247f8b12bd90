// Scheduler.h -- dependency-tracked gate scheduling under the reference's API (addition; internal to libcuHE.so).
//
// The reference ends every public operation with cudaStreamSynchronize (cuhe/CuHE.cu:98,121,139,157) and its clients
// issue one gate at a time from one host thread per device (examples/Prince/Prince.cu:194-322): the GPU runs one small
// kernel at a time although, e.g., the six products and sixteen S-boxes of a PRINCE layer are independent.  In scheduled
// mode (setScheduled(true) or CUHE_SCHED=1 in the environment) a public gate only RECORDS a task: which polynomials it
// reads, which it writes, and the gate itself as a closure over their scheduler-side objects.  A small pool of worker
// threads owned by the library issues the tasks -- every device of multiGPUs(n) has its own workers, queues and lock, each
// worker on its own stream of its device (and, like every host thread of this library, with its own scratch): a task is issued as soon as the tasks it depends on have been issued, ordered on
// the GPU by events (stream-wait on the event of every dependency that ran on another stream).  The client thread blocks
// only where it needs a value on the host (x2z), a raw device pointer, or calls synchronize().
//
// This header is the graph half: tasks, nodes, workers.  CuHE.cpp binds the gates to it.
#pragma once
#include <atomic>
#include <functional>
#include <vector>

namespace cuHE {
class CuPolynomial;
namespace sched {

struct Task;
// scheduler-side state of one client polynomial: the real object (device buffers, host value) lives here while the client
// object only mirrors the metadata.  `obj` is touched by one task at a time (writes are ordered after every earlier task
// on the node, reads after the last write).
struct Node {
	CuPolynomial *obj = nullptr;
	Task *lastWrite = nullptr;             // graph state, guarded by the scheduler's recording lock (only the recording side touches it)
	std::vector<Task *> readers;
	std::atomic<int> refs{1};              // the client object + every task that has not run yet
};

bool on();                                 // scheduled mode is in effect
bool inWorker();                           // the calling thread is one of the scheduler's workers
void *workerStream();                      // inside a task: the stream the task runs on
void start(int threads);                   // idempotent; threads = workers PER DEVICE (<= 0: CUHE_SCHED_THREADS or the default, 3)
void stop();                               // drains, joins the workers, leaves scheduled mode
int threads();

Node *newNode(CuPolynomial *obj);
void releaseNode(Node *n);                 // the client object lets go (tasks may still hold the node)
// record a gate: fn(stream) runs on a worker once every dependency has been issued.
// kind != 0: a BATCHABLE gate on ONE polynomial (`subject`, among the writes): ready tasks of one (kind, key, device) wait in a
// staging area until the workers have nothing else to issue, then up to maxBatch of them run as ONE call of the batch runner
// (gather the subjects' blocks, one array entry point of the C ABI over all their rows, scatter) instead of their closures.
// op1 / op2: the operands of a batchable binary gate (subject = op1 (x) op2)
// kind == kReleaseOnly: fn only RELEASES the device blocks of the polynomial it writes (taskFree) and enqueues nothing.  Such a task
// waits for nothing on its worker's stream: every block it releases carries the positions of the task's dependencies (the block's real
// last uses), so that the next taker on the stream of those uses -- the batch of the next layer -- gets it without any event.
constexpr int kReleaseOnly = -1;
// kind == kHostBlocking (and every task submitted with keep = true): fn ends with a wait on the host -- an upload from a host value, the copy
// down of x2z, each with its synchronise.  Such a task goes to a worker that takes NO groups when the device has one, so that the batches
// of the other client threads do not queue behind a PCIe copy.
constexpr int kHostBlocking = -2;
Task *submit(int dev, const std::vector<Node *> &reads, const std::vector<Node *> &writes, std::function<void(void *)> fn, bool keep = false,
             int kind = 0, long key = 0, Node *subject = nullptr, Node *op1 = nullptr, Node *op2 = nullptr);
typedef void (*BatchRunner)(int kind, Node *const *subjects, Node *const *op1, Node *const *op2, int count, void *stream);
void setBatchRunner(BatchRunner r, int maxBatch);      // (r == nullptr or CUHE_SCHED_BATCH=0: every task runs its own closure)
void wait(Task *t);                        // until t has run on its worker and its device work has finished; drops the reference `keep` took
void waitNode(Node *n);                    // until everything recorded on the node so far has finished on the device
void drain();                              // until everything recorded so far has finished on the device
// device blocks inside a task: taskAlloc returns a block some task released (ordered behind its last use) or a fresh one
// from the library; taskFree keeps a block taskAlloc handed out for the next taker (false: not one of those)
void *taskAlloc(int dev, size_t bytes);
bool taskFree(int dev, void *ptr);
void forgetBlock(void *ptr);               // a block released outside any task
void adoptBlock(int dev, void *ptr, size_t bytes);   // a block taken from the library OUTSIDE any task, idle, whose owner enters the graph: taskFree may keep it
struct Stats { long tasks, crossStreamWaits, maxQueued; };
Stats stats();

} // namespace sched
} // namespace cuHE
