// test_sched_soak.cpp -- the gate scheduler under the conditions a drop-in client can put it in (the evidence for
// making scheduled gates the DEFAULT).  A DHS scheme on the toy ring; every result is checked by decryption against host arithmetic
// mod 2, and bit for bit against the same gates in synchronous mode where a raw pointer is read.
//
//   test_sched_soak [reps]            the soak: `reps` rounds (default 3) of
//       1. operands destroyed while their gates are still queued (scoped temporaries, as Prince.cu:194-322 uses them)
//       2. setScheduled(false) with work queued, results read synchronously, then scheduled again (restart of the workers)
//       3. the raw-pointer getters and x2z keep the reference's observable synchronous semantics (cuhe/CuHE.cu:98,121,139,157):
//          the pointer nRepRead() returns right after a gate was RECORDED holds that gate's result
//       4. two client threads recording on their own objects at the same time
//       5. a second initCuHE on the same ring (a scheme object from a key string, examples/DHS/DHS.cu:57-118) with gates queued
//   test_sched_soak allocfail N       failure injection: the (N+1)-th device allocation after set-up fails; the program must END with
//                                     the reference's error convention (message + exit(-1), cuhe/Debug.h:35-66) -- not hang, not crash
//   (CUHE_SCHED=0 in the environment runs the same program on synchronous gates.)
#include "dhs_client.hpp"
#include "Debug.h"
#include <chrono>
#include <mutex>
#include <thread>
using namespace cuHE;
using dhs_client::Dhs;
typedef long long i64;

static int failures = 0;
static std::mutex reportMu;
static void report(const char *what, bool ok) { std::lock_guard<std::mutex> lk(reportMu); printf("%s\t%s\n", what, ok ? "right" : "wrong"); fflush(stdout); if (!ok) ++failures; }
static ZZX bitsOf(dhs_client::Rng &r, int n) { ZZX z; for (int i = n - 1; i >= 0; --i) if (r.bit()) SetCoeff(z, i, 1); return z; }
static ZZX mulMod2(const ZZX &a, const ZZX &b, const std::vector<i64> &phi) {
	const int n = (int)phi.size() - 1;
	std::vector<i64> t(2 * n, 0);
	for (int i = 0; i < n; ++i) if (!IsZero(coeff(a, i))) for (int j = 0; j < n; ++j) if (!IsZero(coeff(b, j))) ++t[i + j];
	for (int k = 2 * n - 1; k >= n; --k) { const i64 c = t[k]; if (!c) continue; for (int i = 0; i <= n; ++i) t[k - n + i] -= c * phi[i]; }
	ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, to_ZZ((long)(((t[i] % 2) + 2) % 2)));
	return r;
}
static ZZX addMod2(const ZZX &a, const ZZX &b, int n) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, (coeff(a, i) + coeff(b, i)) % to_ZZ(2)); return r; }

struct Inputs { std::vector<ZZX> x, y, cx, cy, andWant, xorWant; };
static Inputs makeInputs(Dhs &dhs, int count, unsigned seed) {
	Inputs in; dhs_client::Rng r(seed);
	for (int i = 0; i < count; ++i) {
		in.x.push_back(bitsOf(r, dhs.n)); in.y.push_back(bitsOf(r, dhs.n));
		in.cx.push_back(dhs.encrypt(in.x[i], 0)); in.cy.push_back(dhs.encrypt(in.y[i], 0));
		in.andWant.push_back(mulMod2(in.x[i], in.y[i], dhs.phi)); in.xorWant.push_back(addMod2(in.x[i], in.y[i], dhs.n));
	}
	return in;
}
// AND + relin + modSwitch and XOR of every pair; the operands live only while their gates are RECORDED
static void recordPairs(const Inputs &in, std::vector<CuCtxt> &prod, std::vector<CuCtxt> &sum) {
	const int count = (int)in.x.size();
	for (int i = 0; i < count; ++i) {
		CuCtxt a, b;                                              // destroyed at the end of this iteration: their gates may not have run yet
		a.setLevel(0, 0, in.cx[i]); b.setLevel(0, 0, in.cy[i]);
		a.x2n(); b.x2n();
		cAnd(prod[i], a, b);
		cXor(sum[i], a, b);
		prod[i].relin(); prod[i].modSwitch();
	}
}
static bool checkPairs(Dhs &dhs, const Inputs &in, std::vector<CuCtxt> &prod, std::vector<CuCtxt> &sum) {
	bool ok = true;
	for (size_t i = 0; i < in.x.size(); ++i) {
		prod[i].x2z(); sum[i].x2z();
		ok = ok && prod[i].level() == 1 && dhs.decrypt(prod[i].zRep(), 1) == in.andWant[i] && dhs.decrypt(sum[i].zRep(), 0) == in.xorWant[i];
	}
	return ok;
}

int main(int argc, char **argv) {
	const bool allocFail = argc > 2 && std::string(argv[1]) == "allocfail";
	const int reps = !allocFail && argc > 1 ? atoi(argv[1]) : 3;
	multiGPUs(1);
	Dhs dhs;
	dhs.setup(3, 2, 8, 40, 20, 1155);
	const bool schedDefault = isScheduled();
	printf("gates are %s after initCuHE (CUHE_SCHED=%s)\n", schedDefault ? "scheduled" : "synchronous", getenv("CUHE_SCHED") ? getenv("CUHE_SCHED") : "unset");
	const int count = 48;
	Inputs in = makeInputs(dhs, count, 99);

	if (allocFail) {
		const long n = atol(argv[2]);
		synchronize();
		cuhe_hip_set_alloc_cache(0);                              // freed blocks go back to the device: every buffer below is a fresh allocation
		cuhe_hip_set_alloc_fail_after(n);
		std::vector<CuCtxt> prod(count), sum(count);
		recordPairs(in, prod, sum);
		const bool ok = checkPairs(dhs, in, prod, sum);
		cuhe_hip_set_alloc_fail_after(-1);
		printf("allocation %ld was never reached (%s)\n", n, ok ? "results right" : "results WRONG");
		return ok ? 3 : 4;                                          // (the test asks for N small enough to be reached: these are failures of the TEST)
	}

	// the same gates on synchronous gates: the bits a raw pointer must show in any mode
	setScheduled(false);
	std::vector<uint64> syncBits;
	{
		CuCtxt a, b, z; a.setLevel(0, 0, in.cx[0]); b.setLevel(0, 0, in.cy[0]);
		a.x2n(); b.x2n(); cAnd(z, a, b);
		syncBits.resize(z.nRepSize() / sizeof(uint64));
		CSC(cuhe_hip_memcpy_d2h(0, syncBits.data(), z.nRepRead(), z.nRepSize(), 0));
		CSC(cuhe_hip_stream_sync(0, 0));
	}
	if (schedDefault) setScheduled(true);

	for (int rep = 0; rep < reps; ++rep) {
		{	// 1
			std::vector<CuCtxt> prod(count), sum(count);
			recordPairs(in, prod, sum);
			report("operands destroyed while their gates are queued", checkPairs(dhs, in, prod, sum));
		}
		{	// 2
			std::vector<CuCtxt> prod(count), sum(count);
			recordPairs(in, prod, sum);
			setScheduled(false);                                    // drains; the objects come back as plain ones when they are touched next
			const bool ok = !isScheduled() && checkPairs(dhs, in, prod, sum);
			if (schedDefault) setScheduled(true);
			std::vector<CuCtxt> prod2(count), sum2(count);
			recordPairs(in, prod2, sum2);
			report("setScheduled(false) with work queued, then scheduled again", ok && isScheduled() == schedDefault && checkPairs(dhs, in, prod2, sum2));
		}
		{	// 3
			CuCtxt a, b, z; a.setLevel(0, 0, in.cx[0]); b.setLevel(0, 0, in.cy[0]);
			a.x2n(); b.x2n(); cAnd(z, a, b);
			std::vector<uint64> bits(z.nRepSize() / sizeof(uint64));
			const uint64 *p = z.nRepRead();                         // right after the gate was recorded
			CSC(cuhe_hip_memcpy_d2h(0, bits.data(), p, z.nRepSize(), 0));
			CSC(cuhe_hip_stream_sync(0, 0));
			bool ok = bits == syncBits;
			CuCtxt w; cXor(w, a, b);
			(void)w.nRepRead();                                     // a pointer went out ...
			cXor(w, w, a);                                          // ... and the object is recorded on again: x + y + x = y
			w.x2z();
			ok = ok && dhs.decrypt(w.zRep(), 0) == in.y[0];
			report("a raw pointer read right after a gate holds the gate's result", ok);
		}
		{	// 4
			Inputs in2 = makeInputs(dhs, count, 1234 + rep);
			bool ok[2] = {false, false};
			std::thread t0([&] { std::vector<CuCtxt> prod(count), sum(count); recordPairs(in, prod, sum); ok[0] = checkPairs(dhs, in, prod, sum); });
			std::thread t1([&] { std::vector<CuCtxt> prod(count), sum(count); recordPairs(in2, prod, sum); ok[1] = checkPairs(dhs, in2, prod, sum); });
			t0.join(); t1.join();
			report("two client threads recording at the same time", ok[0] && ok[1]);
		}
		{	// 5
			std::vector<CuCtxt> prod(count), sum(count);
			recordPairs(in, prod, sum);
			std::vector<ZZ> q(param.depth);
			setParameters(3, 2, 8, 40, 20, 1155);
			initCuHE(q.data(), dhs.phiZ);                           // synchronises, keeps the context
			report("initCuHE on the same ring with gates queued", q[0] == dhs.q[0] && checkPairs(dhs, in, prod, sum));
		}
	}
	setScheduled(false);
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
