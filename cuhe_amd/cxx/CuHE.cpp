// CuHE.cpp -- host side of the public API (CuHE.h) on top of the C ABI.
// Restates the behaviour of cuhe/CuHE.cu, cuhe/Operations.cu (driver half),
// cuhe/Parameters.cu, cuhe/DeviceManager.cu and cuhe/Relinearization.cu of the
// reference: same state machine, same argument checks and messages, but every
// device action is a call into libcuhe_hip.so.
#include "CuHE.h"
#include "Debug.h"
#include "DeviceManager.h"
#include "Operations.h"
#include "Relinearization.h"
#include "Scheduler.h"

#include <cstdio>
#include <cstring>
#include <utility>
#include <exception>
#include <vector>

namespace cuHE {

// ------------------------------------------------------------------ parameters
GlobalParameters param;

static void pullParams() {
	cuhe_params_t q;
	CSC(cuhe_hip_get_parameters(&q));
	param.mSize = q.mSize; param.modLen = q.modLen; param.modLen2 = q.modLen2;
	param.rawLen = q.rawLen; param.crtLen = q.crtLen; param.nttLen = q.nttLen;
	param.logCoeffMax = q.logCoeffMax; param.logCoeffMin = q.logCoeffMin; param.logCoeffCut = q.logCoeffCut;
	param.depth = q.depth; param.modMsg = q.modMsg; param.logMsg = q.logMsg; param.wordsMsg = q.wordsMsg;
	param.logRelin = q.logRelin; param.numEvalKey = q.numEvalKey;
	param.logCrtPrime = q.logCrtPrime; param.numCrtPrime = q.numCrtPrime;
}
void setParam(int d, int p, int w, int min, int cut, int m) {
	CSC(cuhe_hip_set_parameters(d, p, w, min, cut, m));
	pullParams();
}
void resetParam() {
	CSC(cuhe_hip_reset_parameters());
	pullParams();
}
int GlobalParameters::_numCrtPrime(int lvl) {
	if (lvl != -1 && lvl >= depth) {           // cuhe/Parameters.cu:110-113
		cout << "Error: numCrtPrime(lvl) has lvl: " << lvl << endl;
		detail::die(0);
	}
	return cuhe_hip_num_crt_prime(lvl);
}
int GlobalParameters::_logCoeff(int lvl) {
	if (lvl > depth) {                         // cuhe/Parameters.cu:125-128
		cout << "Error: lvl cannot be more than depth!" << endl;
		detail::die(-1);
	}
	return cuhe_hip_log_coeff(lvl);
}
int GlobalParameters::_wordsCoeff(int lvl) { return cuhe_hip_words_coeff(lvl); }
int GlobalParameters::_numEvalKey(int lvl) { return cuhe_hip_num_eval_key(lvl); }
int GlobalParameters::_getLevel(int logq) { return cuhe_hip_get_level(logq); }

// ------------------------------------------------------------------ devices / allocator
static thread_local int tlsDevice = 0;
void setNumDevices(int val) { CSC(cuhe_hip_multi_gpus(val)); }
int numDevices() { return cuhe_hip_num_gpus(); }
static bool allocatorOn = false;
// The reference's allocator takes the device's whole memory when it boots (cuhe/DeviceManager.cu:56-64) and hands out blocks of ONE size.
// Here the pool grows on demand, and booting it reserves a bounded number of blocks of that size on every device -- CUHE_POOL_RESERVE
// blocks, default 512, at most 8 GiB per device -- so that the first operations after startAllocator() do not pay hipMalloc block by block
// (a homomorphic PRINCE block's first run: ~600 blocks, 3-6 ms of idle GPU, profiles/r06_prince_first_block.txt).
void bootDeviceAllocator(size_t blockBytes, unsigned long) {
	CSC(cuhe_hip_start_allocator());
	allocatorOn = true;
	if (blockBytes == 0 || !cuhe_hip_is_initialised()) return;
	const char *e = getenv("CUHE_POOL_RESERVE");
	long want = e ? atol(e) : 512;
	const long cap = (long)(((size_t)8 << 30) / blockBytes);
	if (want > cap) want = cap;
	for (int d = 0; d < numDevices() && want > 0; d++) CSC(cuhe_hip_reserve_blocks(d, blockBytes, (int)want));
}
void haltDeviceAllocator() { CSC(cuhe_hip_stop_allocator()); allocatorOn = false; }
bool deviceAllocatorIsOn() { return allocatorOn; }
void selectDevice(int dev) { tlsDevice = dev; }
int selectedDevice() { return tlsDevice; }
void *deviceMalloc(size_t size) {
	void *p = cuhe_hip_malloc(tlsDevice, size);
	if (!p) CSC(CUHE_EHIP);
	return p;
}
void deviceFree(void *ptr) { CSC(cuhe_hip_free(tlsDevice, ptr)); }
DeviceAllocator::DeviceAllocator() : device_(tlsDevice) {}
DeviceAllocator::~DeviceAllocator() { freeAll(); }
char *DeviceAllocator::allocate(std::ptrdiff_t size) {
	char *p = (char *)cuhe_hip_malloc(device_, (size_t)size);
	if (!p) CSC(CUHE_EHIP);
	allocatedBlocks[p] = size;
	return p;
}
void DeviceAllocator::deallocate(char *ptr) {
	if (allocatedBlocks.erase(ptr)) CSC(cuhe_hip_free(device_, ptr));
}
void DeviceAllocator::freeAll() {
	for (auto &kv : allocatedBlocks) CSC(cuhe_hip_free(device_, kv.first));
	allocatedBlocks.clear();
}

// ---- asynchronous gates (addition).  The reference ends every public operation with a stream synchronise
// (cuhe/CuHE.cu:98,121,...).  With setAsynchronous(true) the operations only ENQUEUE work on their stream: device
// buffers are allocated and released in stream order (cuhe_hip_malloc_stream / free_stream), and the caller
// synchronises when it needs a result on the host (x2z does it itself) or hands a ciphertext to another stream.
static bool asyncGates = false;
void setAsynchronous(bool on) { asyncGates = on; }
bool isAsynchronous() { return asyncGates; }
// A public operation that is a CHAIN of steps on one stream (relin = x2r ; relinearization ; n2c, the x2* conversions,
// modSwitch) synchronises once, at its end: inside a GateScope the steps only enqueue and release their buffers in stream
// order, exactly like asynchronous gates; the operation as a whole keeps the reference's "returns synchronised" contract.
static thread_local int gateDepth = 0;
struct GateScope { GateScope() { ++gateDepth; } ~GateScope() { --gateDepth; } };
static inline bool streamOrdered() { return asyncGates || gateDepth > 0 || sched::inWorker(); }   // a scheduler task only enqueues

// ---- scheduled gates (addition; Scheduler.h).  The client object mirrors the metadata, the recorded tasks run the very
// same gates below on the scheduler-side objects (from a worker thread, where scheduled() is false).
static void runBatchedGates(int kind, sched::Node *const *subjects, sched::Node *const *op1, sched::Node *const *op2, int count, void *stream);
enum { kMaxGateBatch = 128 };                // ready gates of one kind that run as one call of the array entry points
static bool schedChosenByClient = false;      // the client called setScheduled itself: the environment's default no longer applies
static void switchScheduled(bool on, int threads) {
	if (on) { sched::setBatchRunner(runBatchedGates, kMaxGateBatch); sched::start(threads); } else sched::stop();
}
void setScheduled(bool on, int threads) { schedChosenByClient = true; switchScheduled(on, threads); }
bool isScheduled() { return sched::on(); }
void synchronize() { if (sched::on() && !sched::inWorker()) sched::drain(); }
// Scheduled gates are the DEFAULT since round 6 (initCuHE switches them on): an unchanged reference client -- one gate per call from one host
// thread -- then runs a PRINCE block in 0.06 s instead of 0.55 s.  What such a client can observe stays the reference's: x2z(), the raw-pointer
// getters, the setters and synchronize() return with the work done (cuhe/CuHE.cu:98,121,139,157), misuse is reported at the call.
// CUHE_SCHED=0 in the environment (or setScheduled(false)) gives the reference's synchronise-per-gate execution; CUHE_SCHED=n > 1 sets
// the workers per device.  The evidence behind the default: tests/cxx/test_sched_soak.cpp, tools/sched_soak.sh, profiles/r06_sched_soak.txt.
static void schedFromEnvironment() {
	if (schedChosenByClient) return;
	const char *e = getenv("CUHE_SCHED");
	const int v = e ? atoi(e) : 1;
	if (v > 0 && !sched::on()) switchScheduled(true, v > 1 ? v : 0);
	if (v <= 0 && sched::on()) switchScheduled(false, 0);
}
static bool schedCheck() { static const bool on = getenv("CUHE_SCHED_CHECK") && atoi(getenv("CUHE_SCHED_CHECK")) > 0; return on; }
struct SchedAccess {
	static cudaStream_t &stream(CuPolynomial &p) { return p.stream_; }
	static int &level(CuCtxt &c) { return c.level_; }
	static CuCtxt &ct(sched::Node *n) { return *static_cast<CuCtxt *>(n->obj); }
	static CuPtxt &pt(sched::Node *n) { return *static_cast<CuPtxt *>(n->obj); }
	// `dst` becomes what a gate makes of its output when it is shaped like `like` (prepareOut, copy)
	static void runBatch(int kind, sched::Node *const *subjects, sched::Node *const *op1, sched::Node *const *op2, int count, void *stream);
	static void setDevice(CuPolynomial &p, int dev) { p.device_ = dev; }
	static void setProd(CuPolynomial &p, bool prod, int terms) { p.isProd_ = prod; p.prodTerms_ = terms; }
	static void shapeLike(CuCtxt &dst, CuCtxt &like, int domain) {
		dst.level_ = like.level_; dst.logq_ = like.logq_; dst.device_ = like.device_; dst.domain_ = domain;
		dst.isProd_ = false; dst.prodTerms_ = 0; clear(dst.zRep_);
	}
};
typedef std::vector<sched::Node *> Nodes;
#define GATE_SYNC(dev, st) do { if (!streamOrdered()) CSC(cuhe_hip_stream_sync(dev, st)); } while (0)

// (A/B: CUHE_KEEP_CRT=0 frees the CRT rows at c2n as rounds 1-4 did)
static bool keepCrtRows() { static const bool on = !(getenv("CUHE_KEEP_CRT") && atoi(getenv("CUHE_KEEP_CRT")) == 0); return on; }
static void *devAlloc(int dev, size_t bytes, cudaStream_t st = 0) {
	void *p = sched::inWorker() ? sched::taskAlloc(dev, bytes) : streamOrdered() ? cuhe_hip_malloc_stream(dev, bytes, st) : cuhe_hip_malloc(dev, bytes);
	if (!p) CSC(CUHE_EHIP);
	return p;
}

// ------------------------------------------------------------------ Operations.h drivers
static vector<ZZ> coeffModuli;
void getCoeffModuli(ZZ *dst) { for (int i = 0; i < param.depth; i++) dst[i] = coeffModuli[i]; }
// The pre-computation entry points of Operations.h.  initCuHE composes them in the reference (cuhe/CuHE.cu:36-50:
// initNtt ; initCrt ; initBarrett); here the C-ABI library does the whole set-up in cuhe_hip_init, so a client that
// composes them ITSELF gets the same end state: initCrt initialises the library on the ring's cyclotomic polynomial
// if nothing has yet (the CRT primes and coefficient moduli depend on the parameters only), initBarrett installs the
// polynomial modulus it is given (a second initialisation if initCrt had to guess), initNtt has nothing left to do (the
// transform tables are made by cuhe_hip_init and on first use).  The gen* / set* helpers of cuhe/Operations.cu:37-160
// produced host-side tables that live inside the library now: called on an initialised library they are no-ops,
// called before any initialisation they stop the program with a message, like every misuse in this API.
static bool initialisedByParts = false;
static void loadCoeffModuli() {
	coeffModuli.assign(param.depth, ZZ());
	vector<uint8> buf(4096);
	for (int lvl = 0; lvl < param.depth; lvl++) {
		size_t n = 0;
		CSC(cuhe_hip_get_coeff_modulus(lvl, buf.data(), buf.size(), &n));
		coeffModuli[lvl] = ZZFromBytes(buf.data(), (long)n);
	}
}
static void installModulus(const ZZX &modulus) {
	// polynomial modulus as small signed integers, low to high, monic of degree modLen
	vector<int32_t> mod(param.modLen + 1, 0);
	for (int i = 0; i <= param.modLen; i++) {
		ZZ c = coeff(modulus, i);
		long v; conv(v, c);
		mod[i] = (int32_t)v;
	}
	// a second initCuHE on the ring the library already runs on (a second scheme object from a key string, examples/DHS/DHS.cu:57-118)
	// keeps tables, evaluation keys and the blocks live objects hold; another ring starts from scratch
	if (cuhe_hip_is_initialised() && !cuhe_hip_same_ring(mod.data(), (int)mod.size())) CSC(cuhe_hip_shutdown());
	CSC(cuhe_hip_init(mod.data(), (int)mod.size()));
	loadCoeffModuli();
}
static void requireInit(const char *fn) {
	if (!cuhe_hip_is_initialised()) {
		cout << "Error: " << fn << "() called before initCuHE / initCrt: the tables it stands for are made by the library's initialisation." << endl;
		terminate();
	}
}
void initCrt(ZZ *coeffModulus) {
	if (!cuhe_hip_is_initialised()) {
		CSC(cuhe_hip_init(NULL, 0));                   // the m-th cyclotomic polynomial until initBarrett names the modulus
		loadCoeffModuli();
		initialisedByParts = true;
	}
	getCoeffModuli(coeffModulus);
}
void initNtt() {}
void initBarrett(ZZX m) {
	if (!cuhe_hip_is_initialised() || initialisedByParts) installModulus(m);
	initialisedByParts = false;
}
void loadIcrtConst(int, int, cudaStream_t) { requireInit("loadIcrtConst"); }   // every level's constants stay resident (no re-upload, no sync)
void genCrtPrimes() { requireInit("genCrtPrimes"); }
void genCoeffModuli() { requireInit("genCoeffModuli"); }
void genCrtInvPrimes() { requireInit("genCrtInvPrimes"); }
void genIcrtByLevel(int) { requireInit("genIcrtByLevel"); }
void genIcrt() { requireInit("genIcrt"); }
void setPolyModulus(ZZX) { requireInit("setPolyModulus"); }
void createBarrettTemporySpace() { requireInit("createBarrettTemporySpace"); }
uint32 *inttResult(int dev) { return cuhe_hip_intt_result(dev); }
uint64 *ptrNttSwap(int dev) { return (uint64 *)cuhe_hip_ntt_swap(dev); }
uint32 *ptrNttHold(int dev) { return cuhe_hip_intt_result(dev); }
uint64 **ptrNttSwap() {
	static thread_local vector<uint64 *> v;
	v.assign(numDevices(), NULL);
	for (int d = 0; d < (int)v.size(); d++) v[d] = ptrNttSwap(d);
	return v.data();
}
uint32 **ptrNttHold() {
	static thread_local vector<uint32 *> v;
	v.assign(numDevices(), NULL);
	for (int d = 0; d < (int)v.size(); d++) v[d] = ptrNttHold(d);
	return v.data();
}

#define U64P(p) ((uint64_t *)(p))
void crt(uint32 *dst, uint32 *src, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_crt(dst, src, logq, dev, st)); }
void icrt(uint32 *dst, uint32 *src, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_icrt(dst, src, logq, dev, st)); }
void crtAdd(uint32 *sum, uint32 *x, uint32 *y, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_crt_add(sum, x, y, logq, dev, st)); }
void crtAddInt(uint32 *sum, uint32 *x, unsigned a, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_crt_add_int(sum, x, a, logq, dev, st)); }
void crtAddNX1(uint32 *sum, uint32 *x, uint32 *s, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_crt_add_nx1(sum, x, s, logq, dev, st)); }
void crtMulInt(uint32 *prod, uint32 *x, int a, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_crt_mul_int(prod, x, a, logq, dev, st)); }
void crtModSwitch(uint32 *dst, uint32 *src, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_crt_mod_switch(dst, src, logq, dev, st)); }
void _ntt(uint64 *X, uint32 *x, int dev, cudaStream_t st) { CSC(cuhe_hip_ntt_one(U64P(X), x, dev, st)); }
void _nttw(uint64 *X, uint32 *x, int coeffwords, int relinIdx, int dev, cudaStream_t st) { CSC(cuhe_hip_nttw_one(U64P(X), x, coeffwords, relinIdx, dev, st)); }
void _intt(uint32 *x, uint64 *X, int crtidx, int dev, cudaStream_t st) { CSC(cuhe_hip_intt_one(x, U64P(X), crtidx, dev, st)); }
void ntt(uint64 *X, uint32 *x, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_ntt(U64P(X), x, logq, dev, st)); }
void nttw(uint64 *X, uint32 *x, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_nttw(U64P(X), x, logq, dev, st)); }
void intt(uint32 *x, uint64 *X, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_intt(x, U64P(X), logq, dev, st)); }
void inttHold(uint64 *X, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_intt_hold(U64P(X), logq, dev, st)); }
void inttDoubleDeg(uint32 *x, uint64 *X, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_intt_double_deg(x, U64P(X), logq, dev, st)); }
void inttMod(uint32 *x, uint64 *X, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_intt_mod(x, U64P(X), logq, dev, st)); }
void nttMul(uint64 *z, uint64 *y, uint64 *x, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_ntt_mul(U64P(z), U64P(y), U64P(x), logq, dev, st)); }
void nttMulNX1(uint64 *z, uint64 *x, uint64 *s, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_ntt_mul_nx1(U64P(z), U64P(x), U64P(s), logq, dev, st)); }
void nttAdd(uint64 *z, uint64 *y, uint64 *x, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_ntt_add(U64P(z), U64P(y), U64P(x), logq, dev, st)); }
void nttAddNX1(uint64 *z, uint64 *x, uint64 *s, int logq, int dev, cudaStream_t st) { CSC(cuhe_hip_ntt_add_nx1(U64P(z), U64P(x), U64P(s), logq, dev, st)); }
void barrett(uint32 *dst, uint32 *src, int lvl, int dev, cudaStream_t st) { CSC(cuhe_hip_barrett(dst, src, lvl, dev, st)); }
void barrett(uint32 *dst, int lvl, int dev, cudaStream_t st) { CSC(cuhe_hip_barrett_hold(dst, lvl, dev, st)); }

// ------------------------------------------------------------------ relinearisation
void initRelin(ZZX *evalkey) {
	// raw layout at level 0: u32[numEvalKey][rawLen][W0], little-endian words (cuhe/CuHE.cu:324-326)
	const int W0 = param._wordsCoeff(0);
	vector<uint32> host((size_t)param.numEvalKey * param.rawLen * W0, 0);
	for (int k = 0; k < param.numEvalKey; k++)
		for (int i = 0; i < param.rawLen; i++)
			BytesFromZZ((uint8 *)&host[((size_t)k * param.rawLen + i) * W0], coeff(evalkey[k], i), W0 * sizeof(uint32));
	CSC(cuhe_hip_init_relin(host.data()));
}
void saveRelinearization(const char *path) {
	const size_t bytes = cuhe_hip_relin_cache_size();
	vector<uint8> img(bytes);
	CSC(cuhe_hip_relin_export(img.data(), bytes, 0));
	FILE *f = fopen(path, "wb");
	if (!f || fwrite(img.data(), 1, bytes, f) != bytes) { fprintf(stderr, "saveRelinearization: cannot write %s\n", path); exit(-1); }
	fclose(f);
}
bool loadRelinearization(const char *path) {
	FILE *f = fopen(path, "rb");
	if (!f) { fprintf(stderr, "loadRelinearization: cannot open %s\n", path); return false; }
	fseek(f, 0, SEEK_END);
	const long bytes = ftell(f);
	fseek(f, 0, SEEK_SET);
	vector<uint8> img(bytes > 0 ? bytes : 0);
	const bool readOk = bytes > 0 && fread(img.data(), 1, (size_t)bytes, f) == (size_t)bytes;
	fclose(f);
	if (!readOk) { fprintf(stderr, "loadRelinearization: cannot read %s\n", path); return false; }
	if (cuhe_hip_relin_import(img.data(), img.size()) != CUHE_OK) { fprintf(stderr, "loadRelinearization: %s\n", cuhe_hip_last_error()); return false; }
	return true;
}
void relinearization(uint64 *dst, uint32 *src, int lvl, int dev, cudaStream_t st) {
	CSC(cuhe_hip_relinearization(U64P(dst), src, lvl, dev, st));
}

// ------------------------------------------------------------------ library init
void setParameters(int d, int p, int w, int min, int cut, int m) { setParam(d, p, w, min, cut, m); }
void resetParameters() { resetParam(); }
void multiGPUs(int num) { setNumDevices(num); }
int numGPUs() { return numDevices(); }
void startAllocator() { synchronize(); bootDeviceAllocator((size_t)param.numCrtPrime * param.nttLen * sizeof(uint64)); }
void stopAllocator() { synchronize(); haltDeviceAllocator(); }
void initRelinearization(ZZX *evalkey) { synchronize(); initRelin(evalkey); }

void initCuHE(ZZ *coeffMod_, ZZX modulus) {
	synchronize();
	schedFromEnvironment();
	initNtt();
	installModulus(modulus);
	initialisedByParts = false;
	initCrt(coeffMod_);
}

// ------------------------------------------------------------------ CuPolynomial
static void misuse(const char *msg) { cout << msg << endl; terminate(); }

CuPolynomial::CuPolynomial() : logq_(-1), domain_(-1), device_(-1), isProd_(false), prodTerms_(0), rRep_(NULL), cRep_(NULL), nRep_(NULL), cKeep_(NULL), stream_(0), node_(NULL), exposed_(false) { clear(zRep_); }
CuPolynomial::~CuPolynomial() { reset(); }
// ---- attached / detached (scheduled mode)
// mulZZX (and whatever else needs its result on the host at once, on objects nobody else can see) runs its gates DIRECTLY on the calling
// thread in scheduled mode too: recording five tasks for one product and waiting for the last buys nothing and cost 8.5-12.4 ms against
// 3.2 ms for the BASELINE config 3 product (profiles/r06_mulzzx_staging.txt) -- what every encrypt / decrypt / key generation of a DHS
// client pays (examples/DHS/DHS.cu:212-252).
static thread_local int tlsDirectGates = 0;
struct DirectGates { DirectGates() { ++tlsDirectGates; } ~DirectGates() { --tlsDirectGates; } };
bool CuPolynomial::scheduled() {
	if (sched::inWorker()) return false;                        // a recorded gate running on the scheduler-side objects
	if (tlsDirectGates > 0 && !node_) return false;             // (an object that IS attached keeps its place in the graph)
	if (!schedulable()) return false;                           // a client's own subclass: no scheduler-side twin can be made of it
	if (sched::on()) return true;
	if (node_) schedDetach();                                   // the mode was switched off: back to a plain object
	return false;
}
void CuPolynomial::moveStateFrom(CuPolynomial &o) {
	logq_ = o.logq_; domain_ = o.domain_; device_ = o.device_; isProd_ = o.isProd_; prodTerms_ = o.prodTerms_;
	{ using std::swap; clear(zRep_); swap(zRep_, o.zRep_); }
	rRep_ = o.rRep_; cRep_ = o.cRep_; nRep_ = o.nRep_; cKeep_ = o.cKeep_; stream_ = o.stream_;
	o.rRep_ = NULL; o.cRep_ = NULL; o.nRep_ = NULL; o.cKeep_ = NULL;
}
void CuCtxt::moveStateFrom(CuPolynomial &o) { CuPolynomial::moveStateFrom(o); level_ = static_cast<CuCtxt &>(o).level_; }
sched::Node *CuPolynomial::schedAttach() {
	if (node_) return node_;
	// work enqueued by asynchronous gates on this object's stream is not known to the scheduler
	if (asyncGates && device_ >= 0 && (rRep_ || cRep_ || nRep_)) CSC(cuhe_hip_stream_sync(device_, stream_));
	if (exposed_ && device_ >= 0) { CSC(cuhe_hip_device_sync(device_)); exposed_ = false; }
	CuPolynomial *obj = newSameKind();
	obj->moveStateFrom(*this);                                  // the metadata stays here as well: the mirror
	node_ = sched::newNode(obj);
	return node_;
}
void CuPolynomial::schedDetach() {
	if (!node_) return;
	sched::waitNode(node_);                                     // this object's thread is the only one that records on the node
	CuPolynomial *obj = node_->obj;
	if (schedCheck() && (obj->logq_ != logq_ || obj->domain_ != domain_ || obj->device_ != device_ || obj->isProd_ != isProd_ || obj->prodTerms_ != prodTerms_)) {
		printf("Error: scheduler mirror out of step: logq %d/%d domain %d/%d device %d/%d isProd %d/%d terms %d/%d\n", obj->logq_, logq_, obj->domain_, domain_,
		       obj->device_, device_, (int)obj->isProd_, (int)isProd_, obj->prodTerms_, prodTerms_);
		terminate();
	}
	moveStateFrom(*obj);
	sched::Node *n = node_; node_ = NULL;
	sched::releaseNode(n);
}
void CuPolynomial::schedRelease() {
	if (!node_) return;
	sched::Node *n = node_; node_ = NULL;
	// (release-only: reset() enqueues nothing, its blocks go back with the positions of their last uses -- Scheduler.h)
	sched::submit(device_ >= 0 ? device_ : 0, Nodes(), Nodes(1, n), [n](void *s) { n->obj->stream_ = s; n->obj->reset(); }, false, sched::kReleaseOnly);
	sched::releaseNode(n);
}
void CuPolynomial::reset() {
	if (node_ && !sched::inWorker()) { if (sched::on()) schedRelease(); else schedDetach(); }
	clear(zRep_);
	dropKeep();
	if (rRep_ != NULL) rRepFree();
	if (cRep_ != NULL) cRepFree();
	if (nRep_ != NULL) nRepFree();
	isProd_ = false; prodTerms_ = 0; logq_ = -1; domain_ = -1; device_ = -1;
}
// the setters and the raw-pointer getters act on the state itself: an attached object is taken back first
#define DETACHED() do { if (node_ && !sched::inWorker()) schedDetach(); } while (0)
void CuPolynomial::logq(int val) { DETACHED(); logq_ = val; }
void CuPolynomial::domain(int val) { DETACHED(); domain_ = val; }
void CuPolynomial::device(int val) { DETACHED(); device_ = val; }
void CuPolynomial::isProd(bool val) { DETACHED(); isProd_ = val; prodTerms_ = val ? (prodTerms_ > 0 ? prodTerms_ : 1) : 0; }
void CuPolynomial::zRep(ZZX val) { DETACHED(); zRep_ = std::move(val); }
void CuPolynomial::rRep(uint32 *val) { DETACHED(); rRep_ = val; }
void CuPolynomial::cRep(uint32 *val) { DETACHED(); cRep_ = val; }
void CuPolynomial::nRep(uint64 *val) { DETACHED(); dropKeep(); nRep_ = val; }
int CuPolynomial::logq() { return logq_; }
int CuPolynomial::domain() { return domain_; }
int CuPolynomial::device() { return device_; }
bool CuPolynomial::isProd() { return isProd_; }
ZZX CuPolynomial::zRep() { return zRep_; }
void CuPolynomial::swapZRep(ZZX &other) { DETACHED(); using std::swap; swap(zRep_, other); }
// a raw pointer handed to the client in scheduled mode may be written through the Operations.h drivers on any stream:
// the object is marked, and recording on it again first waits for the device (schedAttach)
#define EXPOSED() do { if (!sched::inWorker()) { if (node_) schedDetach(); if (sched::on()) exposed_ = true; } } while (0)
uint32 *CuPolynomial::rRep() { EXPOSED(); return rRep_; }
uint32 *CuPolynomial::cRep() { EXPOSED(); return cRep_; }
uint64 *CuPolynomial::nRep() { EXPOSED(); dropKeep(); return nRep_; }        // (the caller may write through it)
const uint64 *CuPolynomial::nRepRead() { EXPOSED(); return nRep_; }
// (inside a recorded gate the operands that are only read may be shared with gates running on other workers: their
// buffers are ordered by the tasks' events, not by this field; written operands get the task's stream from the task)
void CuPolynomial::stream(cudaStream_t st) { if (sched::inWorker()) return; DETACHED(); stream_ = st; }
cudaStream_t CuPolynomial::stream() { return stream_; }
int CuPolynomial::coeffWords() { return (logq_ + 31) / 32; }
size_t CuPolynomial::rRepSize() { return (size_t)param.rawLen * coeffWords() * sizeof(uint32); }

// with the pooled allocator every representation is one fixed-size block (cuhe/CuHE.cu:469,477,485)
static size_t poolBlock() { return (size_t)param.numCrtPrime * param.nttLen * sizeof(uint64); }
void CuPolynomial::rRepCreate(cudaStream_t st) {
	rRep_ = (uint32 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : rRepSize(), stream_ = st);
	CSC(cuhe_hip_memset_async(device_, rRep_, 0, rRepSize(), st));
}
void CuPolynomial::cRepCreate(cudaStream_t st) {
	cRep_ = (uint32 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : cRepSize(), stream_ = st);
	CSC(cuhe_hip_memset_async(device_, cRep_, 0, cRepSize(), st));
}
void CuPolynomial::nRepCreate(cudaStream_t st) {
	nRep_ = (uint64 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : nRepSize(), stream_ = st);
	CSC(cuhe_hip_memset_async(device_, nRep_, 0, nRepSize(), st));
}
// kernels that produce RAW / CRT rows write the modLen coefficients of the ring; the rest of a row has to read as
// zero, which only needs a fill when the ring is shorter than the row.  NTT-domain rows are always written in full.
static bool shortRing() { return param.modLen < param.crtLen; }
void CuPolynomial::rRepAlloc(cudaStream_t st) {
	rRep_ = (uint32 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : rRepSize(), stream_ = st);
	if (shortRing()) CSC(cuhe_hip_memset_async(device_, rRep_, 0, rRepSize(), st));
}
void CuPolynomial::cRepAlloc(cudaStream_t st) {
	cRep_ = (uint32 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : cRepSize(), stream_ = st);
	if (shortRing()) CSC(cuhe_hip_memset_async(device_, cRep_, 0, cRepSize(), st));
}
void CuPolynomial::nRepAlloc(cudaStream_t st) {
	nRep_ = (uint64 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : nRepSize(), stream_ = st);
}
static void devFree(int dev, void *p, cudaStream_t st) {
	if (sched::inWorker()) { if (sched::taskFree(dev, p)) return; }
	else sched::forgetBlock(p);
	CSC(streamOrdered() ? cuhe_hip_free_stream(dev, p, st) : cuhe_hip_free(dev, p));
}
void CuPolynomial::dropKeep() { if (cKeep_) { devFree(device_, cKeep_, stream_); cKeep_ = NULL; } }
void CuPolynomial::rRepFree() { devFree(device_, rRep_, stream_); rRep_ = NULL; }
void CuPolynomial::cRepFree() { devFree(device_, cRep_, stream_); cRep_ = NULL; }
void CuPolynomial::nRepFree() { devFree(device_, nRep_, stream_); nRep_ = NULL; }

// ---- host staging.  The reference packs coefficient by coefficient into a pageable vector and
// copies that (cuhe/CuHE.cu:317-348).  Here each host thread owns one grow-only PINNED buffer (the examples drive
// one OpenMP thread per device, Prince.cu:194-200), the copy is a single async DMA, the target is not memset first (the copy overwrites all of it) and only the
// modLen coefficients that can be non-zero come back.  (Packing with an OpenMP team was measured and dropped: a
// coefficient is a ~100-byte memcpy, and waking 128 host threads costs two orders of magnitude more than the loop.)
struct PinnedStage {
	void *buf = NULL; size_t cap = 0;
	void *get(size_t bytes) {
		if (bytes > cap) {
			if (buf) cuhe_hip_host_free(buf);
			buf = cuhe_hip_host_alloc(bytes);
			if (!buf) CSC(CUHE_EHIP);
			cap = bytes;
		}
		return buf;
	}
	~PinnedStage() { if (buf) cuhe_hip_host_free(buf); }
};
static thread_local PinnedStage tlsStage;

void CuPolynomial::z2r(cudaStream_t st) {
	if (domain_ != 0) { printf("Error: Not in domain ZZX!\n"); terminate(); }
	rRep_ = (uint32 *)devAlloc(device_, deviceAllocatorIsOn() ? poolBlock() : rRepSize(), stream_ = st);
	const int W = coeffWords();
	const long rows = param.rawLen, top = deg(zRep_);
	const size_t rowBytes = (size_t)W * sizeof(uint32);
	uint8 *host = (uint8 *)tlsStage.get(rRepSize());
	// BytesFromZZ takes |coeff| and zero-pads: inputs are non-negative (cuhe/CuHE.cu:325)
	for (long i = 0; i < rows; i++) {
		if (i <= top) BytesFromZZ(host + (size_t)i * rowBytes, coeff(zRep_, i), (long)rowBytes);
		else memset(host + (size_t)i * rowBytes, 0, rowBytes);
	}
	CSC(cuhe_hip_memcpy_h2d(device_, rRep_, host, rRepSize(), st));
	CSC(cuhe_hip_stream_sync(device_, st));
	clear(zRep_);
	domain_ = 1;
}
void CuPolynomial::r2z(cudaStream_t st) {
	if (domain_ != 1) { printf("Error: Not in domain RAW!\n"); terminate(); }
	const int W = coeffWords();
	const long n = param.modLen;
	const size_t rowBytes = (size_t)W * sizeof(uint32);
	uint8 *host = (uint8 *)tlsStage.get((size_t)n * rowBytes);
	CSC(cuhe_hip_memcpy_d2h(device_, host, rRep_, (size_t)n * rowBytes, st));
	CSC(cuhe_hip_stream_sync(device_, st));
	clear(zRep_);
#ifdef CUHE_MINI_NTL
	zRep_.rep.resize(n);
#else
	zRep_.rep.SetLength(n);
#endif
	for (long i = 0; i < n; i++)
		ZZFromBytes(zRep_.rep[i], host + (size_t)i * rowBytes, (long)rowBytes);
	zRep_.normalize();
	rRepFree();
	domain_ = 0;
}
// Scheduled gates and HOST VALUES.  The scheduler schedules gates; a copy between a ZZX and the device, with its packing or its 16 384 big integers
// to rebuild and its wait for the PCIe transfer, runs on the CLIENT's thread, at the call -- where the reference runs it, and as parallel as the
// client is (Prince.cu:188-322: one OpenMP thread per S-box hands four ZZX in, setLevel(lvl, dev, ZZX), and takes four back, x2z ; zRep).  Recorded
// as tasks (rounds 4-6a) they were serialised on the two or three workers of the device, blocked those workers in stream synchronises while
// batches of the other client threads waited, and handed memory allocated on a worker to the client to free: the literal client structure with 8
// threads took 1.2-1.8 s per block against 0.53 s on synchronous gates (profiles/r06_zzx_state_client.txt).  CUHE_CLIENT_STAGING=0: the tasks (A/B).
static bool clientStaging() { static const bool on = !(getenv("CUHE_CLIENT_STAGING") && atoi(getenv("CUHE_CLIENT_STAGING")) == 0); return on; }
// (scheduled mode, client thread) a host value goes up now: the polynomial enters the graph in the RAW domain
void CuPolynomial::hostValueUp(cudaStream_t st) {
	if (node_) schedDetach();                                  // (attached in the ZZX domain: a recorded copy of a host value)
	DirectGates here;
	z2r(st);
	// (idle after z2r's synchronise; a block taken in stream order -- asynchronous gates on as well -- stays the library's)
	if (!asyncGates) sched::adoptBlock(device_, rRep_, deviceAllocatorIsOn() ? poolBlock() : rRepSize());
}
void CuPolynomial::r2c(cudaStream_t st) {
	if (domain_ != 1) { printf("Error: Not in domain RAW!\n"); terminate(); }
	if (logq_ > param.logCrtPrime) {
		cRepAlloc(st);
		crt(cRep_, rRep_, logq_, device_, st);
		GATE_SYNC(device_, st);
		rRepFree();
	} else {                                                   // one word per coefficient: RAW and CRT coincide
		cRep_ = rRep_; rRep_ = NULL;
	}
	domain_ = 2;
}
void CuPolynomial::c2r(cudaStream_t st) {
	if (domain_ != 2) { printf("Error: Not in domain CRT!\n"); terminate(); }
	if (logq_ > param.logCrtPrime) {
		rRepAlloc(st);
		icrt(rRep_, cRep_, logq_, device_, st);
		GATE_SYNC(device_, st);
		cRepFree();
	} else {
		rRep_ = cRep_; cRep_ = NULL;
	}
	domain_ = 1;
}
void CuPolynomial::c2n(cudaStream_t st) {
	if (domain_ != 2) { printf("Error: Not in domain CRT!\n"); terminate(); }
	nRepAlloc(st);
	// the ciphertext-domain transform: cyclic (the reference's ntt()) on general rings, negacyclic of modLen points
	// when the modulus is x^n + 1 (include/cuhe_hip.h, cuhe_hip_ct_*); every NTT-domain gate below works on either
	CSC(cuhe_hip_ct_ntt(U64P(nRep_), cRep_, logq_, device_, st));
	GATE_SYNC(device_, st);
	dropKeep();
	if (keepCrtRows()) { cKeep_ = cRep_; cRep_ = NULL; } else cRepFree();       // the rows it came from: the way back is free while nobody writes the NTT rows
	domain_ = 3;
}
void CuPolynomial::n2c(cudaStream_t st) {
	if (domain_ != 3) { printf("Error: Not in domain NTT!\n"); terminate(); }
	if (cKeep_ && !isProd_) {                                   // unmodified since c2n: the CRT rows are still there
		stream_ = st;
		cRep_ = cKeep_; cKeep_ = NULL;
		prodTerms_ = 0;
		nRepFree();
		domain_ = 2;
		return;
	}
	dropKeep();
	cRepAlloc(st);
	CSC(cuhe_hip_ct_intt(cRep_, U64P(nRep_), logq_, isProd_ ? 1 : 0, device_, st));    // inttMod for products, intt otherwise
	GATE_SYNC(device_, st);
	isProd_ = false; prodTerms_ = 0;
	nRepFree();
	domain_ = 2;
}
// In scheduled mode a conversion is recorded as ONE task that writes this polynomial; the mirror takes the domain the
// conversion ends in (and loses the product mark wherever the chain passes through n2c).
// (a conversion that starts from a host value -- CUHE_CLIENT_STAGING=0 only -- uploads it and waits for the copy: sched::kHostBlocking keeps it off the
// workers that take groups)
#define RECORD_SELF(call) do { const int kind_ = domain_ == 0 ? sched::kHostBlocking : 0; sched::Node *n_ = schedAttach(); \
	sched::submit(device_, Nodes(), Nodes(1, n_), [n_](void *s) { n_->obj->stream_ = s; n_->obj->call; }, false, kind_); } while (0)
// ... as a BATCHABLE gate (Scheduler.h): ready gates of one kind on ciphertexts of one level / domain / device run as one call of
// the array entry points (batchRunner below).  Ciphertexts only; the key says what the closure would find in the object.
enum { kBatchX2C = 1, kBatchX2N = 2, kBatchRelin = 3, kBatchModSwitch = 4, kBatchAnd = 5, kBatchXor = 6, kBatchCopy = 7, kBatchNot = 8 };
static long batchKey(int level, int domain, bool prod) { return (long)level | (long)domain << 8 | (long)(prod ? 1 : 0) << 12; }
#define RECORD_SELF_BATCH(call, kind, key) do { sched::Node *n_ = schedAttach(); \
	sched::submit(device_, Nodes(), Nodes(1, n_), [n_](void *s) { n_->obj->stream_ = s; n_->obj->call; }, false, kind, key, n_); } while (0)
void CuPolynomial::x2z(cudaStream_t st) {
	if (scheduled()) {
		if (domain_ < 1) return;
		if (!clientStaging()) {
			sched::Node *n = schedAttach();
			sched::wait(sched::submit(device_, Nodes(), Nodes(1, n), [n](void *s) { n->obj->stream_ = s; n->obj->x2z(s); }, true));
			if (domain_ == 3) { isProd_ = false; prodTerms_ = 0; }
			domain_ = 0;
			schedDetach();                                             // a host value lives in the client's object
			return;
		}
		// take the polynomial back (waits for the gates recorded on it) and convert HERE, on the thread that asked (see hostValueUp)
		schedDetach();
		DirectGates here;
		x2z(st);
		return;
	}
	GateScope chain;                                           // r2z ends with the copy to the host and its own synchronise
	if (domain_ == 3) n2c(st);
	if (domain_ == 2) c2r(st);
	if (domain_ == 1) r2z(st);
}
void CuPolynomial::x2r(cudaStream_t st) {
	if (scheduled()) {
		if (domain_ == 1 || domain_ < 0) return;
		if (domain_ == 0 && clientStaging()) { hostValueUp(st); if (domain_ == 1) return; }
		RECORD_SELF(x2r(s));
		if (domain_ == 3) { isProd_ = false; prodTerms_ = 0; }
		domain_ = 1;
		return;
	}
	if (domain_ == 0) { z2r(st); return; }
	{ GateScope chain; if (domain_ == 3) n2c(st); if (domain_ == 2) c2r(st); }
	GATE_SYNC(device_, st);
}
void CuPolynomial::x2c(cudaStream_t st) {
	if (scheduled()) {
		if (domain_ == 2 || domain_ < 0) return;
		if (domain_ == 0 && clientStaging()) { hostValueUp(st); if (domain_ == 2) return; }
		CuCtxt *ct = dynamic_cast<CuCtxt *>(this);
		if (ct && domain_ == 3) RECORD_SELF_BATCH(x2c(s), kBatchX2C, batchKey(ct->level(), 3, isProd_));
		else RECORD_SELF(x2c(s));
		if (domain_ == 3) { isProd_ = false; prodTerms_ = 0; }
		domain_ = 2;
		return;
	}
	if (domain_ == 3) { n2c(st); return; }
	{ GateScope chain; if (domain_ == 0) z2r(st); if (domain_ == 1) r2c(st); }
	GATE_SYNC(device_, st);
}
void CuPolynomial::x2n(cudaStream_t st) {
	if (scheduled()) {
		if (domain_ == 3 || domain_ < 0) return;
		if (domain_ == 0 && clientStaging()) { hostValueUp(st); if (domain_ == 3) return; }
		CuCtxt *ct = dynamic_cast<CuCtxt *>(this);
		if (ct && domain_ == 2) RECORD_SELF_BATCH(x2n(s), kBatchX2N, batchKey(ct->level(), 2, false));
		else RECORD_SELF(x2n(s));
		domain_ = 3;
		return;
	}
	{ GateScope chain; if (domain_ == 0) z2r(st); if (domain_ == 1) r2c(st); if (domain_ == 2) c2n(st); }
	GATE_SYNC(device_, st);
}

// ------------------------------------------------------------------ CuCtxt / CuPtxt
static void createRep(CuPolynomial &p, int domain, cudaStream_t st) {
	if (domain == 1) p.rRepCreate(st);
	else if (domain == 2) p.cRepCreate(st);
	else if (domain == 3) p.nRepCreate(st);
}
void CuCtxt::setLevel(int lvl, int domain, int device, cudaStream_t st) {
	if (scheduled()) {
		schedRelease();                                            // (the reference overwrites the pointers; here the old buffers are released)
		level_ = lvl; logq_ = param._logCoeff(lvl); domain_ = domain; device_ = device;
		if (domain_ == 0) { clear(zRep_); return; }
		sched::Node *n = schedAttach();
		sched::submit(device_, Nodes(), Nodes(1, n), [n, lvl, domain, device](void *s) { SchedAccess::ct(n).setLevel(lvl, domain, device, s); });
		return;
	}
	dropKeep();
	level_ = lvl; logq_ = param._logCoeff(lvl); domain_ = domain; device_ = device;
	if (domain_ == 0) clear(zRep_); else createRep(*this, domain_, st);
}
void CuCtxt::setLevelForOutput(int lvl, int domain, int device, cudaStream_t st) {
	if (scheduled()) {
		schedRelease();
		level_ = lvl; logq_ = param._logCoeff(lvl); domain_ = domain; device_ = device;
		if (domain_ == 0) { clear(zRep_); return; }
		sched::Node *n = schedAttach();
		sched::submit(device_, Nodes(), Nodes(1, n), [n, lvl, domain, device](void *s) { SchedAccess::ct(n).setLevelForOutput(lvl, domain, device, s); });
		return;
	}
	dropKeep();
	level_ = lvl; logq_ = param._logCoeff(lvl); domain_ = domain; device_ = device;
	if (domain_ == 0) clear(zRep_);
	else if (domain_ == 1) rRepAlloc(st);
	else if (domain_ == 2) cRepAlloc(st);
	else if (domain_ == 3) nRepAlloc(st);
}
void CuCtxt::setLevel(int lvl, int device, ZZX val) {
	if (scheduled()) schedRelease();
	dropKeep();
	level_ = lvl; logq_ = param._logCoeff(lvl); domain_ = 0; device_ = device; zRep_ = std::move(val);
}
int CuCtxt::level() { return level_; }
size_t CuCtxt::cRepSize() { return (size_t)param._numCrtPrime(level_) * param.crtLen * sizeof(uint32); }
size_t CuCtxt::nRepSize() { return (size_t)param._numCrtPrime(level_) * cuhe_hip_ct_len() * sizeof(uint64); }   // ct rows: nttLen, or modLen on x^n + 1 rings
void CuCtxt::modSwitch(cudaStream_t st) {
	if (logq_ < param.logCoeffMin + param.logCoeffCut) { printf("Error: Cannot do modSwitch on last level!\n"); terminate(); }
	if (scheduled()) {
		sched::Node *n = schedAttach();
		const bool batchable = domain_ == 2 || domain_ == 3;
		sched::submit(device_, Nodes(), Nodes(1, n), [n](void *s) { SchedAccess::stream(*n->obj) = s; SchedAccess::ct(n).modSwitch(s); }, false,
		              batchable ? kBatchModSwitch : 0, batchKey(level_, domain_, isProd_), n);
		if (domain_ == 3) { isProd_ = false; prodTerms_ = 0; }
		domain_ = 2; logq_ -= param.logCoeffCut; level_++;
		return;
	}
	{ GateScope chain; x2c(st); stream_ = st; crtModSwitch(cRep_, cRep_, logq_, device_, st); }
	GATE_SYNC(device_, st);
	logq_ -= param.logCoeffCut;
	level_++;
}
void CuCtxt::modSwitch(int lvl, cudaStream_t st) {
	if (lvl < level_ || lvl >= param.depth) { printf("Error: ModSwitch to unavailable level!\n"); terminate(); }
	while (level_ < lvl) modSwitch(st);       // (the reference's loop never advances level_: SURVEY A.7)
}
void CuCtxt::relin(cudaStream_t st) {
	if (scheduled()) {
		sched::Node *n = schedAttach();
		const bool batchable = domain_ == 2 || domain_ == 3;
		sched::submit(device_, Nodes(), Nodes(1, n), [n](void *s) { SchedAccess::stream(*n->obj) = s; SchedAccess::ct(n).relin(s); }, false,
		              batchable ? kBatchRelin : 0, batchKey(level_, domain_, isProd_), n);
		domain_ = 2; isProd_ = false; prodTerms_ = 0;
		return;
	}
	{
		// x2r ; relinearization ; n2c (cuhe/CuHE.cu:570-581) -- the last two as ONE call of the library on this stream (cuhe_hip_relin_crt:
		// window transforms, key stream, inverse transforms back to back; the form that overlapped the key stream with the transforms
		// lost its A/B and was removed, profiles/r05_relin_overlap_ab.txt); the ciphertext ends reduced, in the CRT domain, as before
		GateScope chain;
		x2r(st);
		cRepAlloc(st);
		CSC(cuhe_hip_relin_crt(cRep_, rRep_, level_, device_, st));
		rRepFree();
		isProd_ = false; prodTerms_ = 0;
		domain_ = 2;
	}
	GATE_SYNC(device_, st);
}
void CuPtxt::setLogq(int logq, int domain, int device, cudaStream_t st) {
	if (scheduled()) {
		schedRelease();
		logq_ = logq; domain_ = domain; device_ = device;
		if (domain_ == 0) { clear(zRep_); return; }
		sched::Node *n = schedAttach();
		sched::submit(device_, Nodes(), Nodes(1, n), [n, logq, domain, device](void *s) { SchedAccess::pt(n).setLogq(logq, domain, device, s); });
		return;
	}
	dropKeep();
	logq_ = logq; domain_ = domain; device_ = device;
	if (domain_ == 0) clear(zRep_); else createRep(*this, domain_, st);
}
void CuPtxt::setLogq(int logq, int device, ZZX val) { if (scheduled()) schedRelease(); dropKeep(); logq_ = logq; domain_ = 0; device_ = device; zRep_ = std::move(val); }
size_t CuPtxt::cRepSize() { return (size_t)param.crtLen * sizeof(uint32); }
size_t CuPtxt::nRepSize() { return (size_t)cuhe_hip_ct_len() * sizeof(uint64); }

// ------------------------------------------------------------------ gates
// scheduled mode on for this call?  (every operand is asked: an object left attached after the mode was switched off is
// taken back by scheduled())
static bool recordGate(CuPolynomial &a, CuPolynomial &b) { const bool x = a.scheduled(), y = b.scheduled(); return x || y; }
static bool recordGate(CuPolynomial &a, CuPolynomial &b, CuPolynomial &c) { const bool x = recordGate(a, b), y = c.scheduled(); return x || y; }
#define OBJ(n) SchedAccess::ct(n)
void copy(CuCtxt &dst, CuCtxt &src, cudaStream_t st) {
	if (&dst == &src) return;
	if (recordGate(dst, src) && src.domain() > 0) {
		sched::Node *ns = src.schedAttach(), *nd = dst.schedAttach();
		sched::submit(src.device(), Nodes(1, ns), Nodes(1, nd), [ns, nd](void *s) { SchedAccess::stream(*nd->obj) = s; copy(OBJ(nd), OBJ(ns), s); },
		              false, src.domain() >= 2 ? kBatchCopy : 0, batchKey(src.level(), src.domain(), false), nd, ns, NULL);
		const bool prod = src.isProd(); const int terms = src.prodTerms();
		SchedAccess::shapeLike(dst, src, src.domain());
		SchedAccess::setProd(dst, prod, terms);
		return;
	}
	src.stream(st);
	dst.reset();
	dst.setLevelForOutput(src.level(), src.domain(), src.device(), st);
	dst.isProd(src.isProd()); dst.prodTerms(src.prodTerms());
	const int dev = dst.device();
	if (dst.domain() == 0) dst.zRep(src.zRep());
	else if (dst.domain() == 1) CSC(cuhe_hip_memcpy_d2d(dev, dst.rRep(), src.rRep(), dst.rRepSize(), st));
	else if (dst.domain() == 2) CSC(cuhe_hip_memcpy_d2d(dev, dst.cRep(), src.cRep(), dst.cRepSize(), st));
	else if (dst.domain() == 3) CSC(cuhe_hip_memcpy_d2d(dev, dst.nRep(), src.nRepRead(), dst.nRepSize(), st));
	if (dev >= 0) GATE_SYNC(dev, st);
}
static void prepareOut(CuCtxt &out, CuCtxt &like, int domain, cudaStream_t st) {
	if (&out != &like) { out.reset(); out.setLevelForOutput(like.level(), domain, like.device(), st); }
}
void cAnd(CuCtxt &out, CuCtxt &in0, CuCtxt &in1, cudaStream_t st) {
	// the result may be either operand: the reference resets `out` whenever it is not the FIRST one (cuhe/CuHE.cu:108-111), which
	// loses an `out` that is the second; products and sums commute, so that call is taken with its operands exchanged
	if (&out == &in1 && &out != &in0) { cAnd(out, in1, in0, st); return; }
	if (in0.device() != in1.device()) misuse("Error: Multiplication of different devices!");
	if (in0.domain() != 3 || in1.domain() != 3) misuse("Error: Multiplication of non-NTT domain!");
	if (in0.logq() != in1.logq()) misuse("Error: Multiplication of different levels!");
	if (recordGate(out, in0, in1)) {
		sched::Node *n0 = in0.schedAttach(), *n1 = in1.schedAttach(), *no = out.schedAttach();
		sched::submit(in0.device(), Nodes{n0, n1}, Nodes(1, no), [n0, n1, no](void *s) { SchedAccess::stream(*no->obj) = s; cAnd(OBJ(no), OBJ(n0), OBJ(n1), s); },
		              false, kBatchAnd, batchKey(in0.level(), 3, false), no, n0, n1);
		if (&out != &in0) SchedAccess::shapeLike(out, in0, 3);
		SchedAccess::setProd(out, true, 1);
		return;
	}
	prepareOut(out, in0, 3, st);
	in0.stream(st); in1.stream(st); out.stream(st);
	{ const uint64 *x = in0.nRepRead(), *y = in1.nRepRead(); CSC(cuhe_hip_ct_mul(U64P(out.nRep()), U64P(x), U64P(y), out.logq(), out.device(), st)); }
	out.isProd(true); out.prodTerms(1);
	GATE_SYNC(out.device(), st);
}
void cAnd(CuCtxt &out, CuCtxt &inc, CuPtxt &inp, cudaStream_t st) {
	if (inc.device() != inp.device()) misuse("Error: Multiplication of different devices!");
	if (inc.domain() != 3 || inp.domain() != 3) misuse("Error: Multiplication of non-NTT domain!");
	if (recordGate(out, inc, inp)) {
		sched::Node *nc = inc.schedAttach(), *np = inp.schedAttach(), *no = out.schedAttach();
		sched::submit(inc.device(), Nodes{nc, np}, Nodes(1, no), [nc, np, no](void *s) { SchedAccess::stream(*no->obj) = s; cAnd(OBJ(no), OBJ(nc), SchedAccess::pt(np), s); });
		if (&out != &inc) SchedAccess::shapeLike(out, inc, 3);
		SchedAccess::setProd(out, true, 1);
		return;
	}
	prepareOut(out, inc, 3, st);
	inc.stream(st); inp.stream(st); out.stream(st);
	{ const uint64 *x = inc.nRepRead(), *y = inp.nRepRead(); CSC(cuhe_hip_ct_mul_nx1(U64P(out.nRep()), U64P(x), U64P(y), out.logq(), out.device(), st)); }
	out.isProd(true); out.prodTerms(1);
	GATE_SYNC(out.device(), st);
}
void cXor(CuCtxt &out, CuCtxt &in0, CuCtxt &in1, cudaStream_t st) {
	if (&out == &in1 && &out != &in0) { cXor(out, in1, in0, st); return; }        // (see cAnd)
	if (in0.device() != in1.device()) misuse("Error: Addition of different devices!");
	if (recordGate(out, in0, in1)) {
		if (in0.logq() != in1.logq()) misuse("Error: Addition of different levels!");
		const int dom = in0.domain();
		if (!((dom == 2 || dom == 3) && in1.domain() == dom)) misuse("Error: Addition of non-CRT-nor-NTT domain!");
		sched::Node *n0 = in0.schedAttach(), *n1 = in1.schedAttach(), *no = out.schedAttach();
		const int terms = in0.prodTerms() + in1.prodTerms();
		const bool prod = in0.isProd() || in1.isProd(), reduced = dom == 3 && in0.isProd() && in1.isProd() && terms > cuhe_hip_ct_prod_headroom();
		// (a sum that has to reduce its operands first is a chain of gates: not batched)
		sched::submit(in0.device(), Nodes{n0, n1}, Nodes(1, no), [n0, n1, no](void *s) { SchedAccess::stream(*no->obj) = s; cXor(OBJ(no), OBJ(n0), OBJ(n1), s); },
		              false, !reduced ? kBatchXor : 0, batchKey(in0.level(), dom, false), no, n0, n1);
		// the mirror of what the gate below leaves in `out`
		if (&out != &in0) SchedAccess::shapeLike(out, in0, dom);
		if (dom == 3) SchedAccess::setProd(out, !reduced && prod, !reduced && prod ? (terms > 0 ? terms : 1) : 0);
		return;
	}
	in0.stream(st); in1.stream(st);
	if (in0.logq() != in1.logq()) misuse("Error: Addition of different levels!");
	if (in0.domain() == 2 && in1.domain() == 2) {
		prepareOut(out, in0, 2, st);
		crtAdd(out.cRep(), in0.cRep(), in1.cRep(), out.logq(), out.device(), st);
	} else if (in0.domain() == 3 && in1.domain() == 3) {
		// A sum of products stays exact in the NTT domain only while its integer coefficients stay inside the range the
		// inverse transform recovers (cuhe_hip_ct_prod_headroom products: ONE on the largest rings in the negacyclic
		// representation, where 2 n p^2 is just below P).  Beyond that the operands are reduced first (n2c), added as
		// residues and taken back (c2n): the reference's contract "NTT-domain in, NTT-domain out", always exact.
		const int terms = in0.prodTerms() + in1.prodTerms();
		if (in0.isProd() && in1.isProd() && terms > cuhe_hip_ct_prod_headroom()) {
			GateScope chain;
			CuCtxt a, b;
			copy(a, in0, st); copy(b, in1, st);
			a.x2c(st); b.x2c(st);
			crtAdd(a.cRep(), a.cRep(), b.cRep(), a.logq(), a.device(), st);
			a.x2n(st);
			if (&out != &in0 && &out != &in1) { out.reset(); }
			copy(out, a, st);
		} else {
			const bool prod = in0.isProd() || in1.isProd();
			if (&out != &in0) { prepareOut(out, in0, 3, st); }
			{ const uint64 *x = in0.nRepRead(), *y = in1.nRepRead(); CSC(cuhe_hip_ct_add(U64P(out.nRep()), U64P(x), U64P(y), out.logq(), out.device(), st)); }
			out.isProd(prod); out.prodTerms(prod ? (terms > 0 ? terms : 1) : 0);
		}
	} else misuse("Error: Addition of non-CRT-nor-NTT domain!");
	GATE_SYNC(out.device(), st);
}
void cXor(CuCtxt &out, CuCtxt &in0, CuPtxt &in1, cudaStream_t st) {
	if (in0.device() != in1.device()) misuse("Error: Addition of different devices!");
	// the products summed into the result: those of the ciphertext plus those of the plaintext (the count must
	// survive the sum with a plaintext, or a later cXor adds more products than the NTT domain holds exactly)
	const bool prodSum = in0.isProd() || in1.isProd();
	const int termSum = prodSum ? (in0.prodTerms() + in1.prodTerms() > 0 ? in0.prodTerms() + in1.prodTerms() : 1) : 0;
	if (recordGate(out, in0, in1)) {
		const int dom = in0.domain();
		if (!((dom == 2 || dom == 3) && in1.domain() == dom)) misuse("Error: Addition of non-CRT-nor-NTT domain!");
		sched::Node *n0 = in0.schedAttach(), *n1 = in1.schedAttach(), *no = out.schedAttach();
		sched::submit(in0.device(), Nodes{n0, n1}, Nodes(1, no), [n0, n1, no](void *s) { SchedAccess::stream(*no->obj) = s; cXor(OBJ(no), OBJ(n0), SchedAccess::pt(n1), s); });
		if (&out != &in0) SchedAccess::shapeLike(out, in0, dom);
		if (dom == 3) SchedAccess::setProd(out, prodSum, termSum);
		return;
	}
	in0.stream(st); in1.stream(st);
	if (in0.domain() == 2 && in1.domain() == 2) {
		prepareOut(out, in0, 2, st);
		crtAddNX1(out.cRep(), in0.cRep(), in1.cRep(), out.logq(), out.device(), st);
	} else if (in0.domain() == 3 && in1.domain() == 3) {
		if (&out != &in0) prepareOut(out, in0, 3, st);
		{ const uint64 *x = in0.nRepRead(), *y = in1.nRepRead(); CSC(cuhe_hip_ct_add_nx1(U64P(out.nRep()), U64P(x), U64P(y), out.logq(), out.device(), st)); }
		out.isProd(prodSum); out.prodTerms(termSum);
	} else misuse("Error: Addition of non-CRT-nor-NTT domain!");
	GATE_SYNC(out.device(), st);
}
void cNot(CuCtxt &out, CuCtxt &in, cudaStream_t st) {
	if (in.domain() != 2) misuse("Error: cNot of non-CRT domain!");
	if (recordGate(out, in)) {
		sched::Node *ni = in.schedAttach(), *no = out.schedAttach();
		sched::submit(in.device(), Nodes(1, ni), Nodes(1, no), [ni, no](void *s) { SchedAccess::stream(*no->obj) = s; cNot(OBJ(no), OBJ(ni), s); },
		              false, kBatchNot, batchKey(in.level(), 2, false), no, ni, NULL);
		if (&out != &in) { const bool prod = in.isProd(); const int terms = in.prodTerms(); SchedAccess::shapeLike(out, in, 2); SchedAccess::setProd(out, prod, terms); }
		return;
	}
	in.stream(st);
	if (&out != &in) {
		// the reference allocates a zeroed result and only writes the constant term (crt_add_int,
		// cuhe/Base.cu:1096): a value-preserving NOT needs the other coefficients too
		copy(out, in, st);
	}
	crtAddInt(out.cRep(), in.cRep(), (unsigned)param.modMsg - 1, out.logq(), out.device(), st);
	GATE_SYNC(out.device(), st);
}
// The destination block is written by a copy on `st`, a stream of the SOURCE device: it must not be a block that is
// only parked in the order of one of the destination's streams (work enqueued there may still touch it), so it comes
// from the settled pool (cuhe_hip_malloc), never from the stream-ordered one.
// the eager call returns with the copy finished (cuhe/CuHE.cu:217-256).  Inside a recorded task nothing waits on the host: the tasks
// that use the moved ciphertext on the destination device are ordered behind this one by the scheduler's events, the source block is
// released in the order of the copying stream, and the destination block comes from the settled pool
static void moveSync(int srcDev, cudaStream_t st) { if (!sched::inWorker()) CSC(cuhe_hip_stream_sync(srcDev, st)); }
static void *peerAlloc(int dev, size_t bytes) { void *p = cuhe_hip_malloc(dev, bytes); if (!p) CSC(CUHE_EHIP); return p; }
void moveTo(CuCtxt &tar, int dstDev, cudaStream_t st) {
	if (dstDev == tar.device()) return;
	if (tar.scheduled() && tar.domain() > 0) {
		// recorded on a stream of the SOURCE device, like the copy itself; the eager code below synchronises that stream
		// before it returns, so tasks on the destination device find the data in place
		sched::Node *n = tar.schedAttach();
		sched::submit(tar.device(), Nodes(), Nodes(1, n), [n, dstDev](void *s) { SchedAccess::stream(*n->obj) = s; moveTo(OBJ(n), dstDev, s); });
		SchedAccess::setDevice(tar, dstDev);
		return;
	}
	const int srcDev = tar.device();
	if (tar.domain() == 1) {
		void *p = peerAlloc(dstDev, deviceAllocatorIsOn() ? poolBlock() : tar.rRepSize());
		CSC(cuhe_hip_memcpy_peer(p, dstDev, tar.rRep(), srcDev, tar.rRepSize(), st));
		moveSync(srcDev, st);
		tar.rRepFree(); tar.rRep((uint32 *)p);
	} else if (tar.domain() == 2) {
		void *p = peerAlloc(dstDev, deviceAllocatorIsOn() ? poolBlock() : tar.cRepSize());
		CSC(cuhe_hip_memcpy_peer(p, dstDev, tar.cRep(), srcDev, tar.cRepSize(), st));
		moveSync(srcDev, st);
		tar.cRepFree(); tar.cRep((uint32 *)p);
	} else if (tar.domain() == 3) {
		void *p = peerAlloc(dstDev, deviceAllocatorIsOn() ? poolBlock() : tar.nRepSize());
		CSC(cuhe_hip_memcpy_peer(p, dstDev, tar.nRep(), srcDev, tar.nRepSize(), st));
		moveSync(srcDev, st);
		tar.nRepFree(); tar.nRep((uint64 *)p);
	}
	tar.device(dstDev);
}
void copyTo(CuCtxt &dst, CuCtxt &src, int dstDev, cudaStream_t st) {
	copy(dst, src, st);
	moveTo(dst, dstDev, st);
}
void cAndRelinSharded(CuCtxt &out, CuCtxt &in0, CuCtxt &in1, cudaStream_t st) {
	if (in0.device() != in1.device()) misuse("Error: Multiplication of different devices!");
	if (in0.domain() != 3 || in1.domain() != 3) misuse("Error: Multiplication of non-NTT domain!");
	if (in0.logq() != in1.logq()) misuse("Error: Multiplication of different levels!");
	if (&out == &in0 || &out == &in1) misuse("Error: cAndRelinSharded needs a separate result!");
	in0.stream(st); in1.stream(st);
	out.reset();
	out.setLevelForOutput(in0.level(), 2, in0.device(), st);
	CSC(cuhe_hip_mul_relin_sharded_inproc(out.cRep(), U64P(in0.nRepRead()), U64P(in1.nRepRead()), in0.level(), in0.device(), st));
	out.isProd(false);
	GATE_SYNC(out.device(), st);
}

// ------------------------------------------------------------------ batched gates of the scheduler (Scheduler.h)
// `count` >= 2 ready gates of ONE kind on ciphertexts of one level, domain and device, each owning its own blocks: their rows
// are gathered into one array, the array entry point of the C ABI runs once over all of them (count * np rows per launch
// instead of np), the results are scattered into the ciphertexts' blocks, and every object is left exactly as its own gate
// would have left it (the client-side mirrors were updated when the gates were recorded).  Bit-identical to the single gates:
// the array entry points are (tests/cxx/test_cuhe_api.cpp, the array classes' PRINCE).
// Scratch arrays of a batch: three grow-only buffers per worker thread and device (a worker runs one batch at a time, on
// its own stream: the next batch's use is ordered behind this one's), grown geometrically -- a PRINCE block sees batches of
// 2 ... 128 ciphertexts at 25 levels, and every fresh hipMalloc of hundreds of megabytes costs milliseconds.
struct BatchScratch {
	struct Buf { void *p = NULL; size_t cap = 0; };
	std::vector<std::vector<Buf>> perDev;
	unsigned long long generation = 0;
	void *get(int dev, int slot, size_t bytes, void *st) {
		if (generation != cuhe_hip_generation()) { perDev.clear(); generation = cuhe_hip_generation(); }     // the library was shut down since: its blocks went with it
		if ((int)perDev.size() <= dev) perDev.resize(dev + 1, std::vector<Buf>(3));
		Buf &b = perDev[dev][slot];
		if (b.cap < bytes) {
			if (b.p) CSC(cuhe_hip_free_stream(dev, b.p, st));        // (work of this stream may still read it: released in its order)
			const size_t cap = bytes > 2 * b.cap ? bytes : 2 * b.cap;
			b.p = cuhe_hip_malloc(dev, cap);
			if (!b.p) CSC(CUHE_EHIP);
			b.cap = cap;
		}
		return b.p;
	}
	~BatchScratch() {
		if (generation != cuhe_hip_generation() || !cuhe_hip_is_initialised()) return;
		for (size_t d = 0; d < perDev.size(); ++d) for (Buf &b : perDev[d]) if (b.p) cuhe_hip_free((int)d, b.p);
	}
};
static thread_local BatchScratch tlsBatchScratch;
void SchedAccess::runBatch(int kind, sched::Node *const *subjects, sched::Node *const *op1, sched::Node *const *op2, int n, void *st) {
	CuCtxt *c[kMaxGateBatch] = {NULL};
	if (n < 2 || n > kMaxGateBatch) misuse("Error: batch of scheduled gates out of range!");
	for (int i = 0; i < n; ++i) { c[i] = &ct(subjects[i]); c[i]->stream_ = st; }
	void *ptr[kMaxGateBatch];
	if (kind == kBatchCopy) {                                    // out[i] = a copy of a[i] (CRT or NTT domain): one launch for the list
		const void *ps[kMaxGateBatch];
		CuCtxt &first = ct(op1[0]);
		const int dom = first.domain_, dev = first.device_;
		const size_t bytes = dom == 3 ? first.nRepSize() : first.cRepSize();
		for (int i = 0; i < n; ++i) {
			CuCtxt &src = ct(op1[i]);
			c[i]->reset(); c[i]->stream_ = st;
			c[i]->setLevelForOutput(src.level_, dom, dev, st);
			c[i]->isProd_ = src.isProd_; c[i]->prodTerms_ = src.prodTerms_;
			ptr[i] = dom == 3 ? (void *)c[i]->nRep_ : (void *)c[i]->cRep_;
			ps[i] = dom == 3 ? (void *)src.nRep_ : (void *)src.cRep_;
		}
		CSC(cuhe_hip_copy_list(ptr, ps, n, bytes, dev, st));
		return;
	}
	if (kind == kBatchNot) {                                     // out[i] = NOT a[i] (CRT domain): one launch for the list, in place where out[i] is a[i]
		const void *ps[kMaxGateBatch];
		CuCtxt &first = ct(op1[0]);
		const int dev = first.device_, logq = first.logq_;
		for (int i = 0; i < n; ++i) {
			CuCtxt &src = ct(op1[i]);
			if (c[i] != &src) { c[i]->reset(); c[i]->stream_ = st; c[i]->setLevelForOutput(src.level_, 2, dev, st); c[i]->isProd_ = src.isProd_; c[i]->prodTerms_ = src.prodTerms_; }
			ptr[i] = c[i]->cRep_; ps[i] = src.cRep_;
		}
		CSC(cuhe_hip_crt_add_int_list(ptr, ps, (unsigned)param.modMsg - 1, n, logq, dev, st));
		return;
	}
	if (kind == kBatchAnd || kind == kBatchXor) {                // out[i] = a[i] (x) b[i]: one launch for the list
		CuCtxt *a[kMaxGateBatch], *b[kMaxGateBatch];
		const void *pa[kMaxGateBatch], *pb[kMaxGateBatch];
		for (int i = 0; i < n; ++i) { a[i] = &ct(op1[i]); b[i] = &ct(op2[i]); }
		const int dom = a[0]->domain_, dev = a[0]->device_, logq = a[0]->logq_;
		for (int i = 0; i < n; ++i) {
			const bool prod = a[i]->isProd_ || b[i]->isProd_; const int terms = a[i]->prodTerms_ + b[i]->prodTerms_;     // (before `out` is reshaped: it may be a[i])
			if (c[i] != a[i]) { c[i]->reset(); c[i]->stream_ = st; c[i]->setLevelForOutput(a[i]->level_, dom, dev, st); }
			if (kind == kBatchAnd) { c[i]->isProd_ = true; c[i]->prodTerms_ = 1; }
			else if (dom == 3) { c[i]->isProd_ = prod; c[i]->prodTerms_ = prod ? (terms > 0 ? terms : 1) : 0; }
			if (dom == 3) c[i]->dropKeep();                          // (written in place when it is its own first operand)
			ptr[i] = dom == 3 ? (void *)c[i]->nRep_ : (void *)c[i]->cRep_;
			pa[i] = dom == 3 ? (void *)a[i]->nRep_ : (void *)a[i]->cRep_;
			pb[i] = dom == 3 ? (void *)b[i]->nRep_ : (void *)b[i]->cRep_;
		}
		if (dom == 3) CSC(cuhe_hip_ct_binop_list(kind == kBatchAnd ? 1 : 0, ptr, pa, pb, n, logq, dev, st));
		else CSC(cuhe_hip_crt_add_list(ptr, pa, pb, n, logq, dev, st));
		return;
	}
	const int dev = c[0]->device_, lvl = c[0]->level_, np = param._numCrtPrime(lvl);
	bool fromNtt = c[0]->domain_ == 3;
	const bool prod = c[0]->isProd_;
	if (fromNtt && !prod && kind != kBatchX2N) {
		// NTT-domain members that were only READ since their c2n still have the CRT rows they came from (CuPolynomial::cKeep_): when
		// every member has, the way back is free -- no gather, no inverse transform (the a, b, c, d of a PRINCE S-box at modSwitch)
		bool all = true;
		for (int i = 0; i < n; ++i) all = all && c[i]->cKeep_ != NULL;
		if (all) {
			for (int i = 0; i < n; ++i) { c[i]->cRep_ = c[i]->cKeep_; c[i]->cKeep_ = NULL; c[i]->nRepFree(); c[i]->domain_ = 2; c[i]->prodTerms_ = 0; }
			if (kind == kBatchX2C) return;
			fromNtt = false;
		} else for (int i = 0; i < n; ++i) c[i]->dropKeep();
	}
	int cls = 2; while (cls < n) cls *= 2;                       // few scratch sizes (the block cache is keyed by size): level-0 rows, 2 / 4 / .. / 64 ciphertexts
	const size_t cRows = (size_t)np * param.crtLen, nRows = (size_t)np * cuhe_hip_ct_len();
	const size_t cBytes = cRows * sizeof(uint32), nBytes = nRows * sizeof(uint64);
	const size_t cScratch = (size_t)cls * param.numCrtPrime * param.crtLen * sizeof(uint32), nScratch = (size_t)cls * param.numCrtPrime * cuhe_hip_ct_len() * sizeof(uint64);
	static const bool listForms = !(getenv("CUHE_SCHED_LISTS") && atoi(getenv("CUHE_SCHED_LISTS")) == 0);     // (A/B: 0 = gather / array call / scatter everywhere)
	if (kind == kBatchX2N) {                                     // c2n of every ciphertext: one transform call over n * np rows
		if (listForms) {                                          // ... that reads every ciphertext's CRT block and writes its new NTT block: no gather, no scatter
			const void *ps[kMaxGateBatch];
			for (int i = 0; i < n; ++i) { ps[i] = c[i]->cRep_; c[i]->nRepAlloc(st); ptr[i] = c[i]->nRep_; }
			CSC(cuhe_hip_ct_ntt_list((uint64_t *const *)ptr, (const uint32_t *const *)ps, n, lvl, dev, st, NULL));
		} else {
		uint32 *cin = (uint32 *)tlsBatchScratch.get(dev, 0, cScratch, st);
		uint64 *nout = (uint64 *)tlsBatchScratch.get(dev, 1, nScratch, st);
		for (int i = 0; i < n; ++i) ptr[i] = c[i]->cRep_;
		CSC(cuhe_hip_gather_blocks(cin, ptr, n, cBytes, dev, st));
		CSC(cuhe_hip_ntt_rows(U64P(nout), cin, n * np, dev, st));
		for (int i = 0; i < n; ++i) { c[i]->nRepAlloc(st); ptr[i] = c[i]->nRep_; }
		CSC(cuhe_hip_scatter_blocks(ptr, nout, n, nBytes, dev, st));
		}
		for (int i = 0; i < n; ++i) {
			c[i]->dropKeep();
			if (keepCrtRows()) { c[i]->cKeep_ = c[i]->cRep_; c[i]->cRep_ = NULL; } else c[i]->cRepFree();      // (as c2n does)
			c[i]->domain_ = 3;
		}
		return;
	}
	if (kind == kBatchModSwitch && !fromNtt && listForms) {      // CRT-domain ciphertexts switch inside their own blocks: no gather, no scatter, no scratch
		for (int i = 0; i < n; ++i) ptr[i] = c[i]->cRep_;
		CSC(cuhe_hip_crt_mod_switch_list(ptr, ptr, lvl, n, dev, st));
		for (int i = 0; i < n; ++i) { c[i]->logq_ -= param.logCoeffCut; c[i]->level_++; }
		return;
	}
	// the other three start from reduced CRT rows of every ciphertext in one array
	uint32 *rows = (uint32 *)tlsBatchScratch.get(dev, 0, cScratch, st);
	// (kernels that produce CRT rows write the modLen coefficients of the ring: on a ring shorter than the row the rest has to read as zero)
	if (shortRing() && fromNtt) CSC(cuhe_hip_memset_async(dev, rows, 0, n * cBytes, st));
	if (fromNtt) {                                               // n2c: inverse transform (+ reduction modulo the polynomial modulus for products)
		for (int i = 0; i < n; ++i) ptr[i] = c[i]->nRep_;
		if (listForms) CSC(cuhe_hip_ct_intt_list(rows, (const uint64_t *const *)ptr, n, lvl, prod ? 1 : 0, dev, st, NULL));     // (reads the ciphertexts' own NTT blocks)
		else {
			uint64 *nin = (uint64 *)tlsBatchScratch.get(dev, 1, nScratch, st);
			CSC(cuhe_hip_gather_blocks(nin, ptr, n, nBytes, dev, st));
			if (prod) CSC(cuhe_hip_intt_mod_batch(rows, U64P(nin), lvl, n, dev, st));
			else CSC(cuhe_hip_intt_batch(rows, U64P(nin), lvl, n, dev, st));
		}
		for (int i = 0; i < n; ++i) { c[i]->cRepAlloc(st); c[i]->nRepFree(); c[i]->domain_ = 2; c[i]->isProd_ = false; c[i]->prodTerms_ = 0; }
	} else {
		for (int i = 0; i < n; ++i) ptr[i] = c[i]->cRep_;
		CSC(cuhe_hip_gather_blocks(rows, ptr, n, cBytes, dev, st));
	}
	size_t outBytes = cBytes;
	uint32 *result = rows, *next = NULL;
	if (kind == kBatchRelin) {
		CSC(cuhe_hip_relin_batch(rows, rows, lvl, n, dev, st));
		for (int i = 0; i < n; ++i) { c[i]->isProd_ = false; c[i]->prodTerms_ = 0; }
	} else if (kind == kBatchModSwitch && listForms) {             // (from the NTT domain) rows of the array -> the ciphertexts' new blocks: no scatter
		const void *ps[kMaxGateBatch];
		for (int i = 0; i < n; ++i) { ptr[i] = c[i]->cRep_; ps[i] = rows + (size_t)i * cRows; }
		CSC(cuhe_hip_crt_mod_switch_list(ptr, ps, lvl, n, dev, st));
		for (int i = 0; i < n; ++i) { c[i]->logq_ -= param.logCoeffCut; c[i]->level_++; }
		return;
	} else if (kind == kBatchModSwitch) {
		next = (uint32 *)tlsBatchScratch.get(dev, 2, cScratch, st);
		if (shortRing()) CSC(cuhe_hip_memset_async(dev, next, 0, n * cBytes, st));
		CSC(cuhe_hip_crt_mod_switch_batch(next, rows, lvl, n, dev, st));
		result = next; outBytes = (size_t)(np - 1) * param.crtLen * sizeof(uint32);
		for (int i = 0; i < n; ++i) { c[i]->logq_ -= param.logCoeffCut; c[i]->level_++; }
	}
	if (fromNtt || kind != kBatchX2C) {                           // (x2c of CRT-domain ciphertexts is never recorded)
		for (int i = 0; i < n; ++i) ptr[i] = c[i]->cRep_;
		CSC(cuhe_hip_scatter_blocks(ptr, result, n, outBytes, dev, st));
	}
}
static void runBatchedGates(int kind, sched::Node *const *subjects, sched::Node *const *op1, sched::Node *const *op2, int count, void *stream) {
	SchedAccess::runBatch(kind, subjects, op1, op2, count, stream);
}

// ------------------------------------------------------------------ NTL interface
// The by-value parameters are the reference's signature (cuhe/CuHE.h:184); they are moved, not copied again, into
// the operands, and the result is swapped out of the product instead of being copied.
void mulZZX(ZZX &out, ZZX in0, ZZX in1, int lvl, int dev, cudaStream_t st) {
	DirectGates direct;                                         // local operands, result needed on the host: nothing to schedule
	CuCtxt a, b;
	a.setLevel(lvl, dev, std::move(in0));
	b.setLevel(lvl, dev, std::move(in1));
	a.x2n(st);
	b.x2n(st);
	cAnd(a, a, b, st);
	a.x2z(st);
	a.swapZRep(out);
}

void mulZZXBatch(ZZX *x, const ZZX *a, const ZZX *b, int count, int lvl, int dev, cudaStream_t st) {
	if (count <= 0) return;
	const int W = (param._logCoeff(lvl) + 31) / 32;
	const size_t rowBytes = (size_t)W * sizeof(uint32), polyBytes = (size_t)param.rawLen * rowBytes;
	// staging: [a_0 .. a_{count-1} | b_0 .. b_{count-1}] going up, the products coming back into the first half
	uint8 *host = (uint8 *)tlsStage.get(2 * count * polyBytes);
	for (int t = 0; t < 2 * count; t++) {
		const ZZX &src = t < count ? a[t] : b[t - count];
		uint8 *dstp = host + (size_t)t * polyBytes;
		const long top = deg(src);
		for (long i = 0; i < param.rawLen; i++) {
			if (i <= top) BytesFromZZ(dstp + (size_t)i * rowBytes, coeff(src, i), (long)rowBytes);
			else memset(dstp + (size_t)i * rowBytes, 0, rowBytes);
		}
	}
	uint32 *d_in = (uint32 *)devAlloc(dev, 2 * count * polyBytes, st), *d_out = (uint32 *)devAlloc(dev, count * polyBytes, st);
	CSC(cuhe_hip_memcpy_h2d(dev, d_in, host, 2 * count * polyBytes, st));
	CSC(cuhe_hip_mul_raw_batch(d_out, d_in, d_in + (size_t)count * param.rawLen * W, lvl, count, dev, st));
	CSC(cuhe_hip_memcpy_d2h(dev, host, d_out, count * polyBytes, st));
	CSC(cuhe_hip_stream_sync(dev, st));
	devFree(dev, d_in, st); devFree(dev, d_out, st);
	const long n = param.modLen;
	for (int t = 0; t < count; t++) {
		ZZX &dst = x[t];
		clear(dst);
#ifdef CUHE_MINI_NTL
		dst.rep.resize(n);
#else
		dst.rep.SetLength(n);
#endif
		const uint8 *srcp = host + (size_t)t * polyBytes;
		for (long i = 0; i < n; i++) ZZFromBytes(dst.rep[i], srcp + (size_t)i * rowBytes, (long)rowBytes);
		dst.normalize();
	}
}

} // namespace cuHE

// ------------------------------------------------------------------ gates on arrays of ciphertexts (CuHEArray.h)
#include "CuHEArray.h"
namespace cuHE {

// A table may still be read by a gate enqueued on any stream (asynchronous gates): wait for the device before the
// block can be handed out again.  Tables are built once per circuit layout, so the synchronisation is off the hot path.
static void freeIndexTable(int dev, int *p) { if (!p) return; if (streamOrdered()) CSC(cuhe_hip_device_sync(dev)); CSC(cuhe_hip_free(dev, p)); }
CuIndexTable::~CuIndexTable() { freeIndexTable(device_, data_); }
void CuIndexTable::set(const std::vector<int> &values, int device) {
	freeIndexTable(device_, data_); data_ = NULL;
	device_ = device; size_ = values.size();
	data_ = (int *)cuhe_hip_malloc(device_, (size_ ? size_ : 1) * sizeof(int));
	if (!data_) CSC(CUHE_EHIP);
	if (size_) CSC(cuhe_hip_memcpy_h2d(device_, data_, values.data(), size_ * sizeof(int), 0));
	CSC(cuhe_hip_stream_sync(device_, 0));                     // `values` may go out of scope
}

static size_t arrayCtWords(int lvl) { return (size_t)param._numCrtPrime(lvl) * param.crtLen; }     // u32 per CRT ciphertext
static size_t arrayCtElems(int lvl) { return (size_t)param._numCrtPrime(lvl) * cuhe_hip_ct_len(); } // u64 per NTT-domain ciphertext (ct rows)
static void arrayMisuse(const char *msg) { printf("Error: %s\n", msg); terminate(); }

void CuCtxtArray::release() {
	// in the order of the stream that last used the storage (a block parked for stream s is only reissued to stream s
	// until s has been synchronised; releasing on another stream could hand it out while kernels still touch it)
	if (cRep_) { devFree(device_, cRep_, stream_); cRep_ = NULL; }
	if (nRep_) { devFree(device_, nRep_, stream_); nRep_ = NULL; }
	count_ = 0; level_ = -1; domain_ = -1; isProd_ = false; prodTerms_ = 0;
}
void CuCtxtArray::create(int count, int lvl, int domain, int device, cudaStream_t st) {
	if (count < 1 || lvl < 0 || lvl >= param.depth || (domain != 2 && domain != 3)) arrayMisuse("CuCtxtArray::create: bad count, level or domain");
	release();
	count_ = count; level_ = lvl; domain_ = domain; device_ = device; isProd_ = false; prodTerms_ = 0; stream_ = st;
	// storage is sized for level 0 whatever the level: a circuit walks down the levels with arrays of the same few
	// counts, and blocks of the same size come back from the library's block cache instead of hipMalloc / hipFree
	if (domain == 2) cRep_ = (uint32 *)devAlloc(device, count * arrayCtWords(0) * sizeof(uint32), st);
	else nRep_ = (uint64 *)devAlloc(device, count * arrayCtElems(0) * sizeof(uint64), st);
}
uint32 *CuCtxtArray::cRep(int i) { return cRep_ ? cRep_ + (size_t)i * arrayCtWords(level_) : NULL; }
uint64 *CuCtxtArray::nRep(int i) { return nRep_ ? nRep_ + (size_t)i * arrayCtElems(level_) : NULL; }

void CuCtxtArray::put(int i, CuCtxt &src, cudaStream_t st) {
	if (i < 0 || i >= count_ || src.level() != level_ || src.domain() != domain_ || src.device() != device_)
		arrayMisuse("CuCtxtArray::put: index, level, domain or device mismatch");
	touch(st);
	if (domain_ == 2) CSC(cuhe_hip_memcpy_d2d(device_, cRep(i), src.cRep(), src.cRepSize(), st));
	else {
		CSC(cuhe_hip_memcpy_d2d(device_, nRep(i), src.nRep(), src.nRepSize(), st));
		isProd_ = isProd_ || src.isProd();
		if (src.prodTerms() > prodTerms_) prodTerms_ = src.prodTerms();      // the array remembers its largest sum of products
	}
	GATE_SYNC(device_, st);
}
void CuCtxtArray::get(CuCtxt &dst, int i, cudaStream_t st) {
	if (i < 0 || i >= count_) arrayMisuse("CuCtxtArray::get: index out of range");
	touch(st);
	dst.reset();
	dst.setLevelForOutput(level_, domain_, device_, st);
	if (domain_ == 2) CSC(cuhe_hip_memcpy_d2d(device_, dst.cRep(), cRep(i), dst.cRepSize(), st));
	else { CSC(cuhe_hip_memcpy_d2d(device_, dst.nRep(), nRep(i), dst.nRepSize(), st)); dst.isProd(isProd_); dst.prodTerms(isProd_ ? (prodTerms_ > 0 ? prodTerms_ : 1) : 0); }
	GATE_SYNC(device_, st);
}
void CuCtxtArray::x2n(cudaStream_t st) {
	if (domain_ == 3) return;
	if (domain_ != 2) arrayMisuse("CuCtxtArray::x2n: empty array");
	touch(st);
	{
		GateScope chain;
		nRep_ = (uint64 *)devAlloc(device_, count_ * arrayCtElems(0) * sizeof(uint64), st);
		CSC(cuhe_hip_ntt_rows(U64P(nRep_), cRep_, count_ * param._numCrtPrime(level_), device_, st));
		devFree(device_, cRep_, st); cRep_ = NULL;
	}
	domain_ = 3; isProd_ = false; prodTerms_ = 0;
	GATE_SYNC(device_, st);
}
void CuCtxtArray::x2c(cudaStream_t st) {
	if (domain_ == 2) return;
	if (domain_ != 3) arrayMisuse("CuCtxtArray::x2c: empty array");
	touch(st);
	{
		GateScope chain;
		cRep_ = (uint32 *)devAlloc(device_, count_ * arrayCtWords(0) * sizeof(uint32), st);
		if (isProd_) CSC(cuhe_hip_intt_mod_batch(cRep_, U64P(nRep_), level_, count_, device_, st));
		else for (int i = 0; i < count_; ++i) CSC(cuhe_hip_ct_intt(cRep(i), U64P(nRep(i)), param._logCoeff(level_), 0, device_, st));
		devFree(device_, nRep_, st); nRep_ = NULL;
	}
	domain_ = 2; isProd_ = false; prodTerms_ = 0;
	GATE_SYNC(device_, st);
}
void CuCtxtArray::relin(cudaStream_t st) {
	{
		GateScope chain;
		x2c(st);
		touch(st);
		CSC(cuhe_hip_relin_batch(cRep_, cRep_, level_, count_, device_, st));
	}
	GATE_SYNC(device_, st);
}
void CuCtxtArray::modSwitch(cudaStream_t st) {
	if (level_ + 1 >= param.depth) arrayMisuse("Cannot do modSwitch on last level!");
	{
		GateScope chain;
		x2c(st);
		touch(st);
		uint32 *next = (uint32 *)devAlloc(device_, count_ * arrayCtWords(0) * sizeof(uint32), st);
		CSC(cuhe_hip_crt_mod_switch_batch(next, cRep_, level_, count_, device_, st));
		devFree(device_, cRep_, st);
		cRep_ = next;
	}
	level_++;
	GATE_SYNC(device_, st);
}
void concat(CuCtxtArray &dst, const std::vector<CuCtxtArray *> &parts, cudaStream_t st) {
	if (parts.empty() || !parts[0]) arrayMisuse("concat: no parts");
	const int lvl = parts[0]->level(), dom = parts[0]->domain(), dev = parts[0]->device();
	int total = 0, terms = 0; bool prod = false;
	for (CuCtxtArray *p : parts) {
		if (!p || p == &dst || p->level() != lvl || p->domain() != dom || p->device() != dev) arrayMisuse("concat: parts must share level, domain and device");
		total += p->count(); prod = prod || p->isProd(); if (p->prodTerms_ > terms) terms = p->prodTerms_;
	}
	{
		GateScope chain;
		dst.create(total, lvl, dom, dev, st);
		int at = 0;
		for (CuCtxtArray *p : parts) {
			p->touch(st);
			if (dom == 2) CSC(cuhe_hip_memcpy_d2d(dev, dst.cRep(at), p->cRep_, p->count() * arrayCtWords(lvl) * sizeof(uint32), st));
			else CSC(cuhe_hip_memcpy_d2d(dev, dst.nRep(at), p->nRep_, p->count() * arrayCtElems(lvl) * sizeof(uint64), st));
			at += p->count();
		}
		dst.isProd_ = prod; dst.prodTerms_ = terms;
	}
	GATE_SYNC(dev, st);
}
void copy(CuCtxtArray &dst, CuCtxtArray &src, cudaStream_t st) { concat(dst, std::vector<CuCtxtArray *>(1, &src), st); }
void slice(CuCtxtArray &dst, CuCtxtArray &src, int first, int count, cudaStream_t st) {
	if (&dst == &src || first < 0 || count < 1 || first + count > src.count() || (src.domain() != 2 && src.domain() != 3)) arrayMisuse("slice: bad range or empty source");
	{
		GateScope chain;
		dst.create(count, src.level(), src.domain(), src.device(), st);
		src.touch(st);
		if (src.domain() == 2) CSC(cuhe_hip_memcpy_d2d(src.device(), dst.cRep_, src.cRep(first), count * arrayCtWords(src.level()) * sizeof(uint32), st));
		else CSC(cuhe_hip_memcpy_d2d(src.device(), dst.nRep_, src.nRep(first), count * arrayCtElems(src.level()) * sizeof(uint64), st));
		dst.isProd_ = src.isProd_; dst.prodTerms_ = src.prodTerms_;
	}
	GATE_SYNC(src.device(), st);
}
// whole array to another device of this process (cuhe/CuHE.cu:217-256 for one ciphertext): a peer copy on `st`, a
// stream of the SOURCE device, into a settled block of the destination; synchronous like moveTo(CuCtxt&)
void moveTo(CuCtxtArray &arr, int dstDev, cudaStream_t st) {
	if (dstDev == arr.device_ || arr.count_ == 0) { arr.device_ = dstDev; return; }
	const int srcDev = arr.device_;
	const size_t bytes = arr.domain_ == 2 ? arr.count_ * arrayCtWords(0) * sizeof(uint32) : arr.count_ * arrayCtElems(0) * sizeof(uint64);
	const size_t used = arr.domain_ == 2 ? arr.count_ * arrayCtWords(arr.level_) * sizeof(uint32) : arr.count_ * arrayCtElems(arr.level_) * sizeof(uint64);
	void *p = peerAlloc(dstDev, bytes);
	void *from = arr.domain_ == 2 ? (void *)arr.cRep_ : (void *)arr.nRep_;
	CSC(cuhe_hip_memcpy_peer(p, dstDev, from, srcDev, used, st));
	CSC(cuhe_hip_stream_sync(srcDev, st));
	devFree(srcDev, from, st);
	if (arr.domain_ == 2) arr.cRep_ = (uint32 *)p; else arr.nRep_ = (uint64 *)p;
	arr.device_ = dstDev; arr.stream_ = 0;
}
void cAnd(CuCtxtArray &out, CuCtxtArray &in, const CuIndexTable &a, const CuIndexTable &b, cudaStream_t st) {
	if (in.domain() != 3 || a.size() != b.size() || a.size() == 0 || &out == &in) arrayMisuse("cAnd on arrays: operands must be in the NTT domain, index tables of equal length");
	{
		GateScope chain;
		out.create((int)a.size(), in.level(), 3, in.device(), st);
		in.touch(st);
		CSC(cuhe_hip_ntt_mul_pairs(U64P(out.nRep_), U64P(in.nRep_), a.data(), b.data(), (int)a.size(), param._numCrtPrime(in.level()), in.device(), st));
		out.isProd_ = true; out.prodTerms_ = 1;
	}
	GATE_SYNC(in.device(), st);
}
void cXor(CuCtxtArray &out, CuCtxtArray &in0, CuCtxtArray *in1, const CuIndexTable &offsets, const CuIndexTable &list,
          const CuIndexTable &addOne, cudaStream_t st) {
	if (in0.domain() != 2 || (in1 && (in1->domain() != 2 || in1->level() != in0.level())) || offsets.size() < 2 || addOne.size() + 1 != offsets.size()
	    || &out == &in0 || &out == in1)
		arrayMisuse("cXor on arrays: operands must be in the CRT domain at one level, offsets = outputs + 1 entries");
	const int nout = (int)offsets.size() - 1;
	{
		GateScope chain;
		out.create(nout, in0.level(), 2, in0.device(), st);
		in0.touch(st); if (in1) in1->touch(st);
		CSC(cuhe_hip_crt_combine(out.cRep_, in0.cRep_, in0.count(), in1 ? in1->cRep_ : NULL, offsets.data(), list.data(), addOne.data(), nout,
		                         in0.level(), in0.device(), st));
	}
	GATE_SYNC(in0.device(), st);
}

} // namespace cuHE
