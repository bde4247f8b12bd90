// test_prince_arrays_cxx.cpp -- homomorphic PRINCE (BASELINE config 5) layer by layer through the C++ array classes of
// cuhe_amd/cxx/CuHEArray.h (CuCtxtArray, CuIndexTable, cAnd over index pairs, cXor over index lists, relin / modSwitch /
// x2n / x2c on whole arrays, copy / concat).  Same circuit, parameter set and known answers as test_prince_batched.cpp,
// which drives the C ABI directly: six pairwise products per S-box, ab and cd relinearised, four cubic products, one
// relinearisation per output bit, a modulus switch after each multiplicative level (examples/Prince/Prince.cu:204-322);
// 0x9fb51935fc3df524 and the 12 round states (Prince.cu:96,108-145).
//
// Multi-GPU (--devices N): the reference's arrangement (examples/Prince/Prince.cu:194-200: one host thread per GPU,
// the 16 S-boxes of a layer spread over the devices) on arrays: the state array is cut into one slice per device,
// each device's thread moves its slice over (peer copy), evaluates its S-boxes with its own index tables and moves the
// result back; the linear layers run on device 0.  --virtual backs every device by the one physical GPU (tests).
//
// usage: test_prince_arrays_cxx [--no-round-checks] [--async] [--devices N] [--virtual] [--json]
#include "dhs_client.hpp"
#include "prince_common.hpp"
#include "CuHEArray.h"
#include <chrono>
#include <memory>
#include <string>
#include <thread>
using namespace cuHE;
using dhs_client::Dhs;
typedef std::chrono::steady_clock clk;
typedef std::unique_ptr<CuCtxtArray> Arr;

struct Lists { CuIndexTable off, list, one; };                 // term lists of a cXor over arrays
static void setLists(Lists &t, const std::vector<std::vector<int>> &lists, const std::vector<int> &consts, int dev = 0) {
	std::vector<int> o(1, 0), l;
	for (auto &x : lists) { l.insert(l.end(), x.begin(), x.end()); o.push_back((int)l.size()); }
	t.off.set(o, dev); t.list.set(l, dev); t.one.set(consts, dev);
}
// outputs of nb S-boxes over the term array [a..d (4 nb) | ab_i, cd_i (2 nb) | ac ad bc bd (4 nb) | abd acd bcd abc (4 nb)]
static void buildSbox(Lists &t, const Anf &f, int nb, int dev) {
	std::vector<std::vector<int>> lists; std::vector<int> consts;
	for (int i = 0; i < nb; ++i)
		for (int o = 0; o < 4; ++o) {
			std::vector<int> l;
			const int lin[4] = {8, 4, 2, 1}, quad[4] = {10, 9, 6, 5}, cub[4] = {13, 11, 7, 14};
			for (int v = 0; v < 4; ++v) if (f.c[o][lin[v]]) l.push_back(4 * i + v);
			if (f.c[o][12]) l.push_back(4 * nb + 2 * i);
			if (f.c[o][3]) l.push_back(4 * nb + 2 * i + 1);
			for (int v = 0; v < 4; ++v) if (f.c[o][quad[v]]) l.push_back(6 * nb + 4 * i + v);
			for (int v = 0; v < 4; ++v) if (f.c[o][cub[v]]) l.push_back(10 * nb + 4 * i + v);
			if (f.c[o][15]) { printf("unexpected S-box structure\n"); exit(2); }
			lists.push_back(l); consts.push_back(f.c[o][0]);
		}
	setLists(t, lists, consts, dev);
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

// the S-box layer of `nb` S-boxes on one device: index tables resident there, evaluation of a state slice
struct SboxUnit {
	int dev, nb;
	CuIndexTable abA, abB, quA, quB, cuA, cuB;
	Lists fwd, inv;
	SboxUnit(int dev_, int nb_) : dev(dev_), nb(nb_) {
		std::vector<int> a, b;
		for (int i = 0; i < nb; ++i) { a.push_back(4 * i); b.push_back(4 * i + 1); a.push_back(4 * i + 2); b.push_back(4 * i + 3); }
		abA.set(a, dev); abB.set(b, dev);                          // ab_i, cd_i: the products that are relinearised
		a.clear(); b.clear();
		for (int i = 0; i < nb; ++i) { const int pr[4][2] = {{0, 2}, {0, 3}, {1, 2}, {1, 3}}; for (auto &p : pr) { a.push_back(4 * i + p[0]); b.push_back(4 * i + p[1]); } }
		quA.set(a, dev); quB.set(b, dev);                          // ac, ad, bc, bd
		a.clear(); b.clear();
		for (int i = 0; i < nb; ++i) {                             // over [a..d | ab_i, cd_i]: abd, acd, bcd, abc
			const int ab = 4 * nb + 2 * i, cd = 4 * nb + 2 * i + 1;
			a.push_back(ab); b.push_back(4 * i + 3); a.push_back(cd); b.push_back(4 * i);
			a.push_back(cd); b.push_back(4 * i + 1); a.push_back(ab); b.push_back(4 * i + 2);
		}
		cuA.set(a, dev); cuB.set(b, dev);
		int invBox[16]; for (int i = 0; i < 16; ++i) invBox[SBOX[i]] = i;
		buildSbox(fwd, anfOf(SBOX), nb, dev); buildSbox(inv, anfOf(invBox), nb, dev);
	}
	// in: 4 nb ciphertexts, CRT domain, level L, on this device (modified: two levels down); returns the 4 nb outputs at L + 2
	Arr eval(CuCtxtArray &in, bool inverse) {
		const Lists &sb = inverse ? inv : fwd;
		CuCtxtArray Sn, ab, qu;
		copy(Sn, in); Sn.x2n();
		cAnd(ab, Sn, abA, abB); cAnd(qu, Sn, quA, quB);
		Sn.release();
		ab.relin();                                                // x2c + relinearisation of the 2 nb products
		qu.x2c();
		in.modSwitch(); ab.modSwitch(); qu.modSwitch();            // level + 1
		CuCtxtArray lowN, cubic;
		concat(lowN, {&in, &ab}); lowN.x2n();
		cAnd(cubic, lowN, cuA, cuB);
		lowN.release();
		cubic.x2c();
		CuCtxtArray terms;
		concat(terms, {&in, &ab, &qu, &cubic});
		Arr out(new CuCtxtArray);
		cXor(*out, terms, NULL, sb.off, sb.list, sb.one);
		out->relin();
		out->modSwitch();                                          // level + 2
		return out;
	}
};

struct Machine {
	Dhs &dhs;
	Arr S, K;                                      // state (64) and keys [k1 | k0] (128), CRT domain, same level, device 0
	std::vector<std::unique_ptr<SboxUnit>> units;  // one per device
	std::vector<int> firstBox;
	Machine(Dhs &d, int ndev) : dhs(d), S(new CuCtxtArray), K(new CuCtxtArray) {
		for (int t = 0, at = 0; t < ndev; ++t) {
			const int nb = 16 / ndev + (t < 16 % ndev ? 1 : 0);
			firstBox.push_back(at); at += nb;
			units.emplace_back(new SboxUnit(t, nb));
		}
	}
	int level() const { return S->level(); }
	void sboxLayer(bool inverse) {
		const int nd = (int)units.size();
		if (nd == 1) {
			Arr out = units[0]->eval(*S, inverse);
			S.swap(out);
		} else {
			// one host thread per device (Prince.cu:194-200): slice -> peer copy -> S-boxes -> peer copy back
			std::vector<Arr> part(nd), res(nd);
			for (int t = 0; t < nd; ++t) { part[t].reset(new CuCtxtArray); slice(*part[t], *S, 4 * firstBox[t], 4 * units[t]->nb); }
			std::vector<std::thread> th;
			for (int t = 0; t < nd; ++t)
				th.emplace_back([&, t] {
					cuhe_hip_pin_thread_to_device(t);                       // the thread that drives device t launches from the CPUs local to it (include/cuhe_hip.h)
					moveTo(*part[t], t);
					res[t] = units[t]->eval(*part[t], inverse);
					if (isAsynchronous() && cuhe_hip_stream_sync(t, NULL) != 0) { printf("stream sync failed\n"); exit(2); }
					moveTo(*res[t], 0);
				});
			for (auto &x : th) x.join();
			std::vector<CuCtxtArray *> parts;
			for (auto &r : res) parts.push_back(r.get());
			Arr out(new CuCtxtArray);
			concat(*out, parts);
			S.swap(out);
		}
		K->modSwitch(); K->modSwitch();                           // the keys follow the state: level + 2
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
	bool checkRounds = true, async = false, virt = false, json = false;
	int ndev = 1, repeat = 1;
	for (int i = 1; i < argc; ++i) {
		const std::string a = argv[i];
		if (a == "--no-round-checks") checkRounds = false; else if (a == "--async") async = true; else if (a == "--virtual") virt = true;
		else if (a == "--json") json = true; else if (a == "--devices" && i + 1 < argc) ndev = atoi(argv[++i]);
		else if (a == "--repeat" && i + 1 < argc) repeat = atoi(argv[++i]);      // the block N times in one process: from the second on the arrays' device memory is not first-time hipMalloc any more
	}
	if (ndev < 1 || ndev > 16) { printf("--devices 1..16\n"); return 2; }
	const u64x F = ~0ULL, pt = 0, key0 = F, key1 = 0;             // the reference's run (Prince.cu:69-74)
	if (virt) cuhe_hip_set_virtual_devices(1);
	multiGPUs(ndev);
	Dhs dhs;
	dhs.setup(25, 2, 16, 25, 25, 21845);
	if (!getenv("CUHE_SCHED")) setScheduled(false);         // this client issues whole layers itself; CUHE_SCHED=1 in the environment runs it on scheduled gates all the same (tools/sched_soak.sh)
	printf("DHS(25,2,16,25,25,21845): n=%d nttLen=%d primes=%d evalKeys=%d\n", dhs.n, param.nttLen, param.numCrtPrime, param.numEvalKey);
	for (int pass = 0; pass < (repeat > 1 ? repeat : 1); ++pass) {
	Machine M(dhs, ndev);
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
	traceMark();
	const auto t0 = clk::now();
	M.linear(whitenIn);
	for (int i = 1; i <= 5; ++i) { M.sboxLayer(false); check(); M.linear(fwdRound[i]); }
	M.sboxLayer(false); check();
	M.linear(middle);
	M.sboxLayer(true); check();
	for (int i = 6; i <= 10; ++i) { M.linear(keyRc[i]); M.linear(invMix); M.sboxLayer(true); check(); }
	M.linear(whitenOut);
	if (cuhe_hip_stream_sync(0, NULL) != 0) { printf("stream sync failed\n"); return 2; }
	const double encSeconds = std::chrono::duration<double>(clk::now() - t0).count() - paused;
	traceMark();
	bool constant; const u64x got = M.decryptState(constant);
	const u64x want = plainPrince(pt, key0, key1, NULL);
	const bool ok = constant && got == want && want == 0x9fb51935fc3df524ULL && M.level() == 24;
	printf("homomorphic PRINCE: %016llx   expected %016llx   %s\n", got, want, ok ? "right" : "wrong");
	if (!ok) ++failures;
	printf("Prince Encryption: %.3f s on %d %sGPU%s, CuCtxtArray gates, %s (round checks excluded)\n", encSeconds, ndev, virt ? "virtual " : "", ndev > 1 ? "s" : "",
	       async ? "asynchronous" : "synchronous");
	if (json) printf("{\"prince_seconds\": %.4f, \"block\": %d, \"devices\": %d, \"virtual\": %s, \"kat\": \"%016llx\", \"kat_ok\": %s, \"round_states_checked\": %d, \"failures\": %d}\n",
	                 encSeconds, pass, ndev, virt ? "true" : "false", got, ok ? "true" : "false", checkRounds ? layer : 0, failures);
	if (isAsynchronous()) { setAsynchronous(false); }
	}
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
