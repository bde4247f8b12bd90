// test_prince_arrays_cxx.cpp -- homomorphic PRINCE (BASELINE config 5) layer by layer through the C++ array classes of
// cuhe_amd/cxx/CuHEArray.h (CuCtxtArray, CuIndexTable, cAnd over index pairs, cXor over index lists, relin / modSwitch /
// x2n / x2c on whole arrays, copy / concat).  Same circuit, parameter set and known answers as test_prince_batched.cpp,
// which drives the C ABI directly: six pairwise products per S-box, ab and cd relinearised, four cubic products, one
// relinearisation per output bit, a modulus switch after each multiplicative level (examples/Prince/Prince.cu:204-322);
// 0x9fb51935fc3df524 and the 12 round states (Prince.cu:96,108-145).
//
// usage: test_prince_arrays_cxx [--no-round-checks] [--async]
#include "dhs_client.hpp"
#include "prince_common.hpp"
#include "CuHEArray.h"
#include <chrono>
#include <memory>
#include <string>
using namespace cuHE;
using dhs_client::Dhs;
typedef std::chrono::steady_clock clk;
typedef std::unique_ptr<CuCtxtArray> Arr;

struct Lists { CuIndexTable off, list, one; };                 // term lists of a cXor over arrays
static void setLists(Lists &t, const std::vector<std::vector<int>> &lists, const std::vector<int> &consts) {
	std::vector<int> o(1, 0), l;
	for (auto &x : lists) { l.insert(l.end(), x.begin(), x.end()); o.push_back((int)l.size()); }
	t.off.set(o); t.list.set(l); t.one.set(consts);
}
// S-box outputs over the term array [a..d (64) | ab_i, cd_i (32) | ac ad bc bd (64) | abd acd bcd abc (64)]
static void buildSbox(Lists &t, const Anf &f) {
	std::vector<std::vector<int>> lists; std::vector<int> consts;
	for (int i = 0; i < 16; ++i)
		for (int o = 0; o < 4; ++o) {
			std::vector<int> l;
			const int lin[4] = {8, 4, 2, 1}, quad[4] = {10, 9, 6, 5}, cub[4] = {13, 11, 7, 14};
			for (int v = 0; v < 4; ++v) if (f.c[o][lin[v]]) l.push_back(4 * i + v);
			if (f.c[o][12]) l.push_back(64 + 2 * i);
			if (f.c[o][3]) l.push_back(64 + 2 * i + 1);
			for (int v = 0; v < 4; ++v) if (f.c[o][quad[v]]) l.push_back(96 + 4 * i + v);
			for (int v = 0; v < 4; ++v) if (f.c[o][cub[v]]) l.push_back(160 + 4 * i + v);
			if (f.c[o][15]) { printf("unexpected S-box structure\n"); exit(2); }
			lists.push_back(l); consts.push_back(f.c[o][0]);
		}
	setLists(t, lists, consts);
}
// linear layer over [state (64) | k1 (64) | k0 (64)]
static void linearTable(Lists &t, const std::vector<std::vector<int>> &from, u64x rc, bool addK1, const std::vector<std::vector<int>> *keyTerms) {
	std::vector<std::vector<int>> lists(64); std::vector<int> consts(64);
	for (int i = 0; i < 64; ++i) {
		lists[i] = from[i];
		if (addK1) lists[i].push_back(64 + i);
		if (keyTerms) for (int e : (*keyTerms)[i]) lists[i].push_back(64 + e);
		consts[i] = (int)((rc >> (63 - i)) & 1);
	}
	setLists(t, lists, consts);
}

static int failures = 0;

struct Machine {
	Dhs &dhs;
	Arr S, K;                                      // state (64) and keys [k1 | k0] (128), CRT domain, same level
	CuIndexTable abA, abB, quA, quB, cuA, cuB;
	Lists sboxFwd, sboxInv;
	explicit Machine(Dhs &d) : dhs(d), S(new CuCtxtArray), K(new CuCtxtArray) {
		std::vector<int> a, b;
		for (int i = 0; i < 16; ++i) { a.push_back(4 * i); b.push_back(4 * i + 1); a.push_back(4 * i + 2); b.push_back(4 * i + 3); }
		abA.set(a); abB.set(b);                                    // ab_i, cd_i: the products that are relinearised
		a.clear(); b.clear();
		for (int i = 0; i < 16; ++i) { const int pr[4][2] = {{0, 2}, {0, 3}, {1, 2}, {1, 3}}; for (auto &p : pr) { a.push_back(4 * i + p[0]); b.push_back(4 * i + p[1]); } }
		quA.set(a); quB.set(b);                                    // ac, ad, bc, bd
		a.clear(); b.clear();
		for (int i = 0; i < 16; ++i) {                             // over [a..d | ab_i, cd_i]: abd, acd, bcd, abc
			const int ab = 64 + 2 * i, cd = 64 + 2 * i + 1;
			a.push_back(ab); b.push_back(4 * i + 3); a.push_back(cd); b.push_back(4 * i);
			a.push_back(cd); b.push_back(4 * i + 1); a.push_back(ab); b.push_back(4 * i + 2);
		}
		cuA.set(a); cuB.set(b);
		int inv[16]; for (int i = 0; i < 16; ++i) inv[SBOX[i]] = i;
		buildSbox(sboxFwd, anfOf(SBOX)); buildSbox(sboxInv, anfOf(inv));
	}
	int level() const { return S->level(); }
	void sboxLayer(const Lists &sb) {
		CuCtxtArray Sn, ab, qu;
		copy(Sn, *S); Sn.x2n();
		cAnd(ab, Sn, abA, abB); cAnd(qu, Sn, quA, quB);
		Sn.release();
		ab.relin();                                                // x2c + relinearisation of the 32 products
		qu.x2c();
		S->modSwitch(); ab.modSwitch(); qu.modSwitch(); K->modSwitch();      // level + 1
		CuCtxtArray lowN, cubic;
		concat(lowN, {S.get(), &ab}); lowN.x2n();
		cAnd(cubic, lowN, cuA, cuB);
		lowN.release();
		cubic.x2c();
		CuCtxtArray terms;
		concat(terms, {S.get(), &ab, &qu, &cubic});
		Arr out(new CuCtxtArray);
		cXor(*out, terms, NULL, sb.off, sb.list, sb.one);
		out->relin();
		out->modSwitch(); K->modSwitch();                         // level + 2
		S.swap(out);
	}
	void linear(const Lists &t) {
		Arr out(new CuCtxtArray);
		cXor(*out, *S, K.get(), t.off, t.list, t.one);
		S.swap(out);
	}
	u64x decryptState(bool &constant) {
		u64x v = 0; constant = true;
		for (int i = 0; i < 64; ++i) {
			CuCtxt t; S->get(t, i); t.x2z();
			const ZZX m = dhs.decrypt(t.zRep(), level());
			constant = constant && deg(m) <= 0;
			v = (v << 1) | (u64x)(IsZero(coeff(m, 0)) ? 0 : 1);
		}
		return v;
	}
};

int main(int argc, char **argv) {
	bool checkRounds = true, async = false;
	for (int i = 1; i < argc; ++i) { if (std::string(argv[i]) == "--no-round-checks") checkRounds = false; else if (std::string(argv[i]) == "--async") async = true; }
	const u64x F = ~0ULL, pt = 0, key0 = F, key1 = 0;             // the reference's run (Prince.cu:69-74)
	multiGPUs(1);
	Dhs dhs;
	dhs.setup(25, 2, 16, 25, 25, 21845);
	printf("DHS(25,2,16,25,25,21845): n=%d nttLen=%d primes=%d evalKeys=%d\n", dhs.n, param.nttLen, param.numCrtPrime, param.numEvalKey);
	Machine M(dhs);
	std::vector<u64x> expect;
	plainPrince(pt, key0, key1, &expect);
	M.S->create(64, 0, 2); M.K->create(128, 0, 2);
	for (int i = 0; i < 192; ++i) {
		const int bit = i < 64 ? (int)((pt >> (63 - i)) & 1) : i < 128 ? (int)((key1 >> (127 - i)) & 1) : (int)((key0 >> (191 - i)) & 1);
		CuCtxt c; c.setLevel(0, 0, dhs.encryptBit(bit, 0)); c.x2c();
		if (i < 64) M.S->put(i, c); else M.K->put(i - 64, c);
	}
	std::vector<std::vector<int>> ident(64), mp = mPrimeSources(), mpSr(64), srInvMp(64);
	for (int i = 0; i < 64; ++i) ident[i] = {i};
	for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) mpSr[4 * i + k] = mp[4 * SR[i] + k];
	{
		int srInvSrc[64];
		for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) srInvSrc[4 * SR[i] + k] = 4 * i + k;
		for (int i = 0; i < 64; ++i) for (int s : mp[i]) srInvMp[i].push_back(srInvSrc[s]);
	}
	std::vector<std::vector<int>> k0terms(64), k0p(64);
	for (int i = 0; i < 64; ++i) { k0terms[i] = {64 + i}; k0p[i] = {64 + (i + 63) % 64}; }
	k0p[63].push_back(64 + 0);
	Lists whitenIn, middle, invMix, whitenOut, fwdRound[6], keyRc[11];
	linearTable(whitenIn, ident, RC[0], true, &k0terms);
	for (int i = 1; i <= 5; ++i) linearTable(fwdRound[i], mpSr, RC[i], true, NULL);
	linearTable(middle, mp, 0, false, NULL); linearTable(invMix, srInvMp, 0, false, NULL);
	for (int i = 6; i <= 10; ++i) linearTable(keyRc[i], ident, RC[i], true, NULL);
	linearTable(whitenOut, ident, RC[11], true, &k0p);
	double paused = 0; int layer = 0;
	auto check = [&] {
		if (checkRounds) {
			if (cuhe_hip_stream_sync(0, NULL) != 0) { printf("stream sync failed\n"); exit(2); }     // the layer's work belongs to the timed part
			const auto c0 = clk::now();
			bool constant; const u64x got = M.decryptState(constant);
			const bool ok = constant && got == expect[layer];
			printf("S-box layer %2d  level %2d  %016llx  %s\n", layer, M.level(), got, ok ? "right" : "wrong");
			if (!ok) ++failures;
			paused += std::chrono::duration<double>(clk::now() - c0).count();
		}
		++layer;
	};
	setAsynchronous(async);                         // the gates only enqueue; get() + x2z() and the end of the timing synchronise
	const auto t0 = clk::now();
	M.linear(whitenIn);
	for (int i = 1; i <= 5; ++i) { M.sboxLayer(M.sboxFwd); check(); M.linear(fwdRound[i]); }
	M.sboxLayer(M.sboxFwd); check();
	M.linear(middle);
	M.sboxLayer(M.sboxInv); check();
	for (int i = 6; i <= 10; ++i) { M.linear(keyRc[i]); M.linear(invMix); M.sboxLayer(M.sboxInv); check(); }
	M.linear(whitenOut);
	if (cuhe_hip_stream_sync(0, NULL) != 0) { printf("stream sync failed\n"); return 2; }
	const double encSeconds = std::chrono::duration<double>(clk::now() - t0).count() - paused;
	bool constant; const u64x got = M.decryptState(constant);
	const u64x want = plainPrince(pt, key0, key1, NULL);
	const bool ok = constant && got == want && want == 0x9fb51935fc3df524ULL && M.level() == 24;
	printf("homomorphic PRINCE: %016llx   expected %016llx   %s\n", got, want, ok ? "right" : "wrong");
	if (!ok) ++failures;
	printf("Prince Encryption: %.3f s on 1 GPU, CuCtxtArray gates, %s (round checks excluded)\n", encSeconds, async ? "asynchronous" : "synchronous");
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
