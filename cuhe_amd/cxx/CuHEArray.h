// CuHEArray.h -- gates on ARRAYS of ciphertexts (addition; the reference's gates take one ciphertext per call,
// cuhe/CuHE.cu:101-215,545-581).  A CuCtxtArray holds `count` ciphertexts of one level contiguously in device memory,
// in the CRT domain (u32[count][np][crtLen]) or the NTT domain (u64[count][np][nttLen]); one call applies a gate to
// the whole array, so that every launch carries count * np rows.  Each operation is bit-identical to the CuCtxt gates
// it stands for (tests/cxx/test_cuhe_api.cpp checks them against each other).  Built on the array entry points of
// include/cuhe_hip.h; synchronisation follows the CuCtxt gates (one stream synchronise per operation unless
// setAsynchronous(true)).
#pragma once
#include "CuHE.h"
#include <vector>

namespace cuHE {

// a table of ciphertext indices in device memory (operand pairs of cAnd, term lists of cXor)
class CuIndexTable {
public:
	CuIndexTable() : data_(NULL), size_(0), device_(0) {}   // (tables are uploaded synchronously on stream 0 and only read afterwards)
	~CuIndexTable();
	void set(const std::vector<int> &values, int device = 0);
	const int *data() const { return data_; }
	size_t size() const { return size_; }
private:
	CuIndexTable(const CuIndexTable &);
	CuIndexTable &operator=(const CuIndexTable &);
	int *data_; size_t size_; int device_;
};

class CuCtxtArray {
public:
	CuCtxtArray() : count_(0), level_(-1), domain_(-1), device_(0), isProd_(false), prodTerms_(0), cRep_(NULL), nRep_(NULL), stream_(0) {}
	~CuCtxtArray() { release(); }
	// `count` ciphertexts of level `lvl` in `domain` (2 = CRT, 3 = NTT), contents undefined
	void create(int count, int lvl, int domain, int device = 0, cudaStream_t st = 0);
	void release();
	int count() const { return count_; }
	int level() const { return level_; }
	int domain() const { return domain_; }
	int device() const { return device_; }
	bool isProd() const { return isProd_; }
	int prodTerms() const { return prodTerms_; }      // the largest number of products summed into one of its ciphertexts (NTT domain)
	uint32 *cRep(int i = 0);          // ciphertext i, CRT domain
	uint64 *nRep(int i = 0);          // ciphertext i, NTT domain
	// copy one ciphertext in / out; the CuCtxt must be in this array's domain, level and device
	void put(int i, CuCtxt &src, cudaStream_t st = 0);
	void get(CuCtxt &dst, int i, cudaStream_t st = 0);
	void x2n(cudaStream_t st = 0);                 // CRT -> NTT of every ciphertext
	void x2c(cudaStream_t st = 0);                 // NTT -> CRT (with the reduction modulo the polynomial modulus for products)
	void relin(cudaStream_t st = 0);               // CuCtxt::relin of every ciphertext; leaves the CRT domain
	void modSwitch(cudaStream_t st = 0);           // CuCtxt::modSwitch of every ciphertext: one level down (CRT domain)
private:
	CuCtxtArray(const CuCtxtArray &);
	CuCtxtArray &operator=(const CuCtxtArray &);
	friend void copy(CuCtxtArray &, CuCtxtArray &, cudaStream_t);
	friend void slice(CuCtxtArray &, CuCtxtArray &, int, int, cudaStream_t);
	friend void moveTo(CuCtxtArray &, int, cudaStream_t);
	friend void concat(CuCtxtArray &, const std::vector<CuCtxtArray *> &, cudaStream_t);
	friend void cAnd(CuCtxtArray &, CuCtxtArray &, const CuIndexTable &, const CuIndexTable &, cudaStream_t);
	friend void cXor(CuCtxtArray &, CuCtxtArray &, CuCtxtArray *, const CuIndexTable &, const CuIndexTable &, const CuIndexTable &, cudaStream_t);
	int count_, level_, domain_, device_;
	bool isProd_;
	int prodTerms_;
	uint32 *cRep_;
	uint64 *nRep_;
	cudaStream_t stream_;             // the stream that last produced or consumed the storage: blocks are released in its order
	void touch(cudaStream_t st) { stream_ = st; }
};

// dst = a copy of src; dst = the ciphertexts of all parts in order (parts of one level, domain and device)
void copy(CuCtxtArray &dst, CuCtxtArray &src, cudaStream_t st = 0);
void concat(CuCtxtArray &dst, const std::vector<CuCtxtArray *> &parts, cudaStream_t st = 0);
// dst = the ciphertexts [first, first + count) of src; the whole array to another device of this process (peer copy)
void slice(CuCtxtArray &dst, CuCtxtArray &src, int first, int count, cudaStream_t st = 0);
void moveTo(CuCtxtArray &arr, int dstDev, cudaStream_t st = 0);
// out[t] = in[a[t]] * in[b[t]]  (NTT domain; `out` is created with a.size() ciphertexts; relin / x2c follow as for cAnd)
void cAnd(CuCtxtArray &out, CuCtxtArray &in, const CuIndexTable &a, const CuIndexTable &b, cudaStream_t st = 0);
// out[o] = sum of the ciphertexts listed in list[offsets[o] .. offsets[o+1]) (+ 1 on the constant coefficient where
// addOne[o] != 0: cNot); entries e < in0.count() address in0[e], the others in1[e - in0.count()].  CRT domain.
void cXor(CuCtxtArray &out, CuCtxtArray &in0, CuCtxtArray *in1, const CuIndexTable &offsets, const CuIndexTable &list,
          const CuIndexTable &addOne, cudaStream_t st = 0);

} // namespace cuHE
