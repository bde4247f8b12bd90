// CuHE.h -- public API of the library, source compatible with cuhe/CuHE.h:41-210
// of vernamlab/cuHE: the DHS / Prince example sources include this header and
// call exactly these symbols.  Implementation: CuHE.cpp on top of the C ABI in
// include/cuhe_hip.h (MI355X / gfx950 kernels); no CUDA headers are needed --
// cudaStream_t is an opaque handle that carries a hipStream_t.
#pragma once
#include "Parameters.h"
#include <NTL/ZZ.h>
#include <NTL/ZZX.h>
NTL_CLIENT

#ifndef CUHE_STREAM_T
#define CUHE_STREAM_T
typedef void *cudaStream_t;
#endif

typedef unsigned char uint8;
typedef unsigned int uint32;
typedef unsigned long int uint64;

namespace cuHE {
namespace sched { struct Node; }
struct SchedAccess;

// A polynomial living in exactly one of four domains:
//   0 = ZZX (host), 1 = RAW (device, big coefficients), 2 = CRT (device,
//   residues per prime), 3 = NTT (device, transform per prime).
// It owns at most one device representation at a time; conversions create the
// target representation and release the source one.
class CuPolynomial {
public:
	CuPolynomial();
	virtual ~CuPolynomial();
	void reset();                    // idempotent; safe after an explicit destructor call
	// setters (pointer setters copy the pointer, not the data)
	void logq(int val);
	void domain(int val);
	void device(int val);
	void isProd(bool val);
	void zRep(ZZX val);
	void rRep(uint32 *val);
	void cRep(uint32 *val);
	void nRep(uint64 *val);
	// getters
	int logq();
	int domain();
	int device();
	bool isProd();
	int prodTerms() { return prodTerms_; }
	void prodTerms(int n) { prodTerms_ = n; }
	ZZX zRep();
	void swapZRep(ZZX &other);       // (addition) exchange the host value without copying it
	uint32 *rRep();
	uint32 *cRep();
	uint64 *nRep();
	const uint64 *nRepRead();        // (addition) the NTT-domain rows for READING: unlike nRep() it leaves the kept CRT rows (below) alone
	void stream(cudaStream_t st);    // (addition) the stream whose work last touched the device representation;
	cudaStream_t stream();           //            buffers are released in that stream's order in asynchronous mode
	// any domain -> the named domain
	void x2z(cudaStream_t st = 0);
	void x2r(cudaStream_t st = 0);
	void x2c(cudaStream_t st = 0);
	void x2n(cudaStream_t st = 0);
	// storage
	void rRepCreate(cudaStream_t st = 0);
	void cRepCreate(cudaStream_t st = 0);
	void nRepCreate(cudaStream_t st = 0);
	void rRepFree();
	void cRepFree();
	void nRepFree();
	int coeffWords();
	size_t rRepSize();
	virtual size_t cRepSize() = 0;
	virtual size_t nRepSize() = 0;
	// (additions) storage for a representation every element of which is about to be written: no memset, except
	// for the tail beyond modLen of RAW/CRT rows when the ring is shorter than the row (kernels write modLen entries)
	void rRepAlloc(cudaStream_t st = 0);
	void cRepAlloc(cudaStream_t st = 0);
	void nRepAlloc(cudaStream_t st = 0);
	// (additions) scheduled mode (setScheduled, Scheduler.h): while a polynomial is ATTACHED its device buffers and host
	// value live in a scheduler-side object (node_->obj) that the recorded gates work on; this object only mirrors the
	// metadata (domain, level, logq, device, isProd).  schedDetach waits for the gates recorded on it and takes the state back.
	sched::Node *schedAttach();
	void schedDetach();
	bool scheduled();                // true: record the operation instead of running it (detaches first when the mode is off)
protected:
	friend struct SchedAccess;
	// (scheduled mode) a fresh object of the same dynamic type for the scheduler's side.  Not pure: the reference's class has no such
	// member (cuhe/CuHE.h:45-110), so a client's own subclass of CuPolynomial keeps compiling -- it simply is not schedulable and
	// its conversions run at once on the calling thread, in scheduled mode too.
	virtual CuPolynomial *newSameKind() const { return NULL; }
	virtual bool schedulable() const { return false; }
	virtual void moveStateFrom(CuPolynomial &other);
	void schedRelease();             // let go of the node: its buffers are released by a recorded task
	void z2r(cudaStream_t st = 0);   // ZZX -> RAW
	void r2z(cudaStream_t st = 0);   // RAW -> ZZX
	void hostValueUp(cudaStream_t st);   // (addition, scheduled mode) ZZX -> RAW at once, on the calling (client) thread
	void r2c(cudaStream_t st = 0);   // CRT
	void c2r(cudaStream_t st = 0);   // ICRT
	void c2n(cudaStream_t st = 0);   // NTT
	void n2c(cudaStream_t st = 0);   // INTT (+ reduction mod the polynomial modulus for products)
	int logq_;
	int domain_;
	int device_;
	bool isProd_;
	int prodTerms_;                  // products summed into an NTT-domain value since it was last reduced (cXor keeps the sum exact)
	ZZX zRep_;
	uint32 *rRep_;
	uint32 *cRep_;
	uint64 *nRep_;
	// (addition) the CRT rows this polynomial was transformed FROM (c2n), kept while its NTT-domain rows stay unmodified: going back
	// (n2c of a non-product: x2c, modSwitch, relin of an operand that was only read) then costs nothing.  Not part of the state a client
	// sees: cRep() is NULL in the NTT domain as in the reference; dropped by everything that writes or hands out the NTT-domain rows.
	uint32 *cKeep_;
	void dropKeep();
	cudaStream_t stream_;
	sched::Node *node_;
	bool exposed_;                   // a raw device pointer was handed out in scheduled mode
};

// ciphertext: a polynomial per CRT prime of its level
class CuCtxt : public CuPolynomial {
public:
	CuCtxt() : CuPolynomial() { level_ = -1; }
	void setLevel(int lvl, int dom, int dev, cudaStream_t st = 0);   // allocate, no value
	void setLevel(int lvl, int dev, ZZX val);                        // host value
	void setLevelForOutput(int lvl, int dom, int dev, cudaStream_t st = 0);   // (addition) like setLevel(lvl, dom, dev) without zero fill
	int level();
	void modSwitch(cudaStream_t st = 0);           // one level down
	void modSwitch(int lvl, cudaStream_t st = 0);  // down to level lvl
	void relin(cudaStream_t st = 0);
	size_t cRepSize();
	size_t nRepSize();
protected:
	friend struct SchedAccess;
	CuPolynomial *newSameKind() const { return new CuCtxt; }
	bool schedulable() const { return true; }
	void moveStateFrom(CuPolynomial &other);
	int level_;
};

// batched plaintext: a single polynomial
class CuPtxt : public CuPolynomial {
public:
	void setLogq(int logq, int dom, int dev, cudaStream_t st = 0);
	void setLogq(int logq, int dev, ZZX val);
	size_t cRepSize();
	size_t nRepSize();
protected:
	CuPolynomial *newSameKind() const { return new CuPtxt; }
	bool schedulable() const { return true; }
};

// initialisation: setParameters first, then (optionally) multiGPUs, then initCuHE.
// initCuHE writes the `depth` coefficient moduli into coeffMod_.
void initCuHE(ZZ *coeffMod_, ZZX modulus);
void startAllocator();
void stopAllocator();
void multiGPUs(int num);
int numGPUs();
void setParameters(int d, int p, int w, int min, int cut, int m);
// (addition) asynchronous gates: with setAsynchronous(true) conversions and gates only enqueue work on their stream
// (the reference synchronises after each one, cuhe/CuHE.cu:98,121,...); the caller synchronises -- cuhe_hip_stream_sync
// or any x2z() -- before it reads a result on the host or hands a ciphertext to work on another stream.  Default: off.
void setAsynchronous(bool on);
bool isAsynchronous();
// (addition) scheduled gates: the SAME client code -- one host thread, default stream, a gate per call -- with the
// independent gates running concurrently.  Every public gate and conversion records a task (what it reads, what it
// writes); worker threads of the library issue the tasks on their own streams as their inputs become available, ordered
// on the GPU by events; ready gates of one kind on ciphertexts of one level run as one call of the array entry points.
// The client blocks only in x2z(), in the raw-pointer getters, and in synchronize().  Results are those of the
// synchronous gates, bit for bit.  Also switched on by CUHE_SCHED=1 (CUHE_SCHED_THREADS=n workers) in the
// environment at initCuHE, so that an unchanged client gets it.  Default: off (the reference's semantics).
void setScheduled(bool on, int threads = 0);
bool isScheduled();
void synchronize();               // everything recorded so far has finished on the device (no-op when not scheduled)
void resetParameters();
void initRelinearization(ZZX *evalkey);

// x = a * b mod (polynomial modulus, q_lvl), host in / host out
void mulZZX(ZZX &x, ZZX a, ZZX b, int lvl, int dev, cudaStream_t st = 0);
// (addition) count independent products x[i] = a[i] * b[i] in one call: one packed upload, every device stage once over
// all count * numCrtPrime rows (cuhe_hip_mul_raw_batch), one download.  Same results as count calls of mulZZX.
void mulZZXBatch(ZZX *x, const ZZX *a, const ZZX *b, int count, int lvl, int dev, cudaStream_t st = 0);

// gates
void copy(CuCtxt &x, CuCtxt &a, cudaStream_t st = 0);
void cAnd(CuCtxt &x, CuCtxt &a, CuCtxt &b, cudaStream_t st = 0);
void cAnd(CuCtxt &x, CuCtxt &c, CuPtxt &p, cudaStream_t st = 0);
void cXor(CuCtxt &x, CuCtxt &a, CuCtxt &b, cudaStream_t st = 0);
void cXor(CuCtxt &x, CuCtxt &c, CuPtxt &p, cudaStream_t st = 0);
void cNot(CuCtxt &x, CuCtxt &a, cudaStream_t st = 0);
void moveTo(CuCtxt &x, int dstDev, cudaStream_t st = 0);
void copyTo(CuCtxt &dst, CuCtxt &src, int dstDev, cudaStream_t st = 0);
// (addition) x = relin(a * b) with the CRT primes of the level split over all numGPUs() devices of this process: a and b
// in the NTT domain on one device, x comes back reduced in the CRT domain on the same device.  Each device multiplies,
// transforms and key-switches its own block of primes; the one exchange is the all-gather of CRT rows before ICRT (peer
// copies over xGMI).  Same result as cAnd(x, a, b); x.relin().  One ciphertext's latency, not throughput: independent
// ciphertexts are better spread over the devices whole (moveTo / the dev argument), as the reference does.
void cAndRelinSharded(CuCtxt &x, CuCtxt &a, CuCtxt &b, cudaStream_t st = 0);

} // namespace cuHE
