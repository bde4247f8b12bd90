// test_prince_batched.cpp -- homomorphic PRINCE (BASELINE config 5) evaluated LAYER BY LAYER on arrays of
// ciphertexts: the 64 state bits (and the 128 key bits) live in contiguous device arrays, and one S-box layer is ~12
// calls of the array gates of include/cuhe_hip.h (cuhe_hip_ntt_rows, ntt_mul_pairs, intt_mod_batch, relin_batch,
// crt_mod_switch_batch, crt_combine) instead of ~4 000 single-ciphertext gates -- the MI355X-first shape of the same
// circuit: every launch carries 64-224 ciphertexts x up to 25 primes of rows.  Same circuit as the reference's
// (examples/Prince/Prince.cu:204-322: six pairwise products per S-box, ab and cd relinearised, four cubic products,
// one relinearisation per output bit, a modulus switch after each multiplicative level), same parameter set, same
// known answers: 0x9fb51935fc3df524 and the 12 round states (Prince.cu:96,108-145).  Keys and ciphertexts come from
// the scheme client of dhs_client.hpp (through CuHE.h); everything between encryption and decryption runs on the arrays.
//
// usage: test_prince_batched [--no-round-checks]
#include "dhs_client.hpp"
#include "prince_common.hpp"
#include <chrono>
#include <string>
using namespace cuHE;
using dhs_client::Dhs;
typedef std::chrono::steady_clock clk;

#define CK(call) do { if ((call) != 0) { printf("%s failed: %s\n", #call, cuhe_hip_last_error()); exit(2); } } while (0)

struct DevInts {            // a small int32 table in device memory
	int32_t *p = NULL; size_t n = 0;
	void set(const std::vector<int> &v) {
		if (p) cuhe_hip_free(0, p);
		n = v.size();
		p = (int32_t *)cuhe_hip_malloc(0, std::max<size_t>(n, 1) * 4);
		if (n) CK(cuhe_hip_memcpy_h2d(0, p, v.data(), n * 4, NULL));
		CK(cuhe_hip_stream_sync(0, NULL));
	}
};
struct Csr { DevInts off, list, add; int nout = 0;
	void set(const std::vector<std::vector<int>> &lists, const std::vector<int> &consts) {
		std::vector<int> o(1, 0), l;
		for (auto &x : lists) { l.insert(l.end(), x.begin(), x.end()); o.push_back((int)l.size()); }
		off.set(o); list.set(l); add.set(consts); nout = (int)lists.size();
	}
};

static int failures = 0;

struct Machine {
	Dhs &dhs;
	int cl, L;
	uint32_t *S, *S2, *K, *K2, *Q, *T, *O;       // CRT-domain arrays
	uint64_t *N, *P;                            // NTT-domain arrays
	int level = 0;
	DevInts qa, qb, ca, cb;
	Csr sboxFwd, sboxInv;
	explicit Machine(Dhs &d) : dhs(d) {
		cl = param.crtLen; L = param.nttLen;
		const size_t np = param.numCrtPrime, ct32 = np * cl * 4, ct64 = np * L * 8;
		S = (uint32_t *)cuhe_hip_malloc(0, 64 * ct32); S2 = (uint32_t *)cuhe_hip_malloc(0, 64 * ct32);
		K = (uint32_t *)cuhe_hip_malloc(0, 128 * ct32); K2 = (uint32_t *)cuhe_hip_malloc(0, 128 * ct32);
		Q = (uint32_t *)cuhe_hip_malloc(0, 96 * ct32); T = (uint32_t *)cuhe_hip_malloc(0, 224 * ct32); O = (uint32_t *)cuhe_hip_malloc(0, 64 * ct32);
		N = (uint64_t *)cuhe_hip_malloc(0, 96 * ct64); P = (uint64_t *)cuhe_hip_malloc(0, 96 * ct64);
		// pair tables: quadratic products first the 32 that are relinearised (ab_i, cd_i), then ac, ad, bc, bd
		std::vector<int> a, b;
		for (int i = 0; i < 16; ++i) { a.push_back(4 * i); b.push_back(4 * i + 1); a.push_back(4 * i + 2); b.push_back(4 * i + 3); }
		for (int i = 0; i < 16; ++i) {
			const int pr[4][2] = {{0, 2}, {0, 3}, {1, 2}, {1, 3}};
			for (auto &p : pr) { a.push_back(4 * i + p[0]); b.push_back(4 * i + p[1]); }
		}
		qa.set(a); qb.set(b);
		// cubic products over [a..d (64) | ab_i, cd_i (32)]: abd, acd, bcd, abc
		a.clear(); b.clear();
		for (int i = 0; i < 16; ++i) {
			const int ab = 64 + 2 * i, cd = 64 + 2 * i + 1;
			a.push_back(ab); b.push_back(4 * i + 3);
			a.push_back(cd); b.push_back(4 * i);
			a.push_back(cd); b.push_back(4 * i + 1);
			a.push_back(ab); b.push_back(4 * i + 2);
		}
		ca.set(a); cb.set(b);
		int inv[16]; for (int i = 0; i < 16; ++i) inv[SBOX[i]] = i;
		buildSbox(sboxFwd, anfOf(SBOX));
		buildSbox(sboxInv, anfOf(inv));
	}
	// term array T at the level after the first multiplications: [0,64) a..d, [64,96) ab_i cd_i, [96,160) ac ad bc bd, [160,224) cubic
	static void buildSbox(Csr &csr, const Anf &f) {
		std::vector<std::vector<int>> lists; std::vector<int> consts;
		for (int i = 0; i < 16; ++i)
			for (int o = 0; o < 4; ++o) {
				std::vector<int> l;
				const int lin[4] = {8, 4, 2, 1};
				for (int v = 0; v < 4; ++v) if (f.c[o][lin[v]]) l.push_back(4 * i + v);
				if (f.c[o][12]) l.push_back(64 + 2 * i);
				if (f.c[o][3]) l.push_back(64 + 2 * i + 1);
				const int quad[4] = {10, 9, 6, 5};
				for (int v = 0; v < 4; ++v) if (f.c[o][quad[v]]) l.push_back(96 + 4 * i + v);
				const int cub[4] = {13, 11, 7, 14};
				for (int v = 0; v < 4; ++v) if (f.c[o][cub[v]]) l.push_back(160 + 4 * i + v);
				if (f.c[o][15]) { printf("unexpected S-box structure\n"); exit(2); }
				lists.push_back(l); consts.push_back(f.c[o][0]);
			}
		csr.set(lists, consts);
	}
	size_t ct32(int lvl) const { return (size_t)param._numCrtPrime(lvl) * cl; }        // words per ciphertext
	size_t ct64(int lvl) const { return (size_t)param._numCrtPrime(lvl) * L; }

	void sboxLayer(const Csr &sb) {
		const int l0 = level, l1 = level + 1, np0 = param._numCrtPrime(l0), np1 = param._numCrtPrime(l1);
		CK(cuhe_hip_ntt_rows(N, S, 64 * np0, 0, NULL));
		CK(cuhe_hip_ntt_mul_pairs(P, N, qa.p, qb.p, 96, np0, 0, NULL));
		CK(cuhe_hip_intt_mod_batch(Q, P, l0, 96, 0, NULL));
		CK(cuhe_hip_relin_batch(Q, Q, l0, 32, 0, NULL));                          // ab_i, cd_i
		CK(cuhe_hip_crt_mod_switch_batch(T, S, l0, 64, 0, NULL));                 // a..d       -> level l1
		CK(cuhe_hip_crt_mod_switch_batch(T + 64 * ct32(l1), Q, l0, 96, 0, NULL)); // quadratics -> level l1
		CK(cuhe_hip_crt_mod_switch_batch(K2, K, l0, 128, 0, NULL));               // keys follow the state's level
		CK(cuhe_hip_ntt_rows(N, T, 96 * np1, 0, NULL));                           // a..d, ab_i, cd_i
		CK(cuhe_hip_ntt_mul_pairs(P, N, ca.p, cb.p, 64, np1, 0, NULL));
		CK(cuhe_hip_intt_mod_batch(T + 160 * ct32(l1), P, l1, 64, 0, NULL));      // cubic terms
		CK(cuhe_hip_crt_combine(O, T, 224, NULL, sb.off.p, sb.list.p, sb.add.p, 64, l1, 0, NULL));
		CK(cuhe_hip_relin_batch(O, O, l1, 64, 0, NULL));
		CK(cuhe_hip_crt_mod_switch_batch(S, O, l1, 64, 0, NULL));                 // -> level l0 + 2
		CK(cuhe_hip_crt_mod_switch_batch(K, K2, l1, 128, 0, NULL));
		level += 2;
	}
	// table of a linear layer: state[i] = sum of state[src] for src in from[i]  +  k1[i] if addK1  +  extra key terms
	// +  constant bit.  Entries >= 64 address the key array K = [k1 | k0].
	static Csr linearTable(const std::vector<std::vector<int>> &from, u64x rc, bool addK1, const std::vector<std::vector<int>> *keyTerms) {
		std::vector<std::vector<int>> lists(64); std::vector<int> consts(64);
		for (int i = 0; i < 64; ++i) {
			lists[i] = from[i];
			if (addK1) lists[i].push_back(64 + i);                                 // K[0..63] = k1
			if (keyTerms) for (int e : (*keyTerms)[i]) lists[i].push_back(64 + e);
			consts[i] = (int)((rc >> (63 - i)) & 1);
		}
		Csr csr; csr.set(lists, consts);
		return csr;
	}
	void linear(const Csr &csr) {
		CK(cuhe_hip_crt_combine(S2, S, 64, K, csr.off.p, csr.list.p, csr.add.p, 64, level, 0, NULL));
		std::swap(S, S2);
	}
	u64x decryptState(bool &constant) {
		CK(cuhe_hip_stream_sync(0, NULL));
		u64x v = 0; constant = true;
		for (int i = 0; i < 64; ++i) {
			CuCtxt t; t.setLevel(level, 2, 0);
			CK(cuhe_hip_memcpy_d2d(0, t.cRep(), S + (size_t)i * ct32(level), ct32(level) * 4, NULL));
			t.x2z();
			const ZZX m = dhs.decrypt(t.zRep(), level);
			constant = constant && deg(m) <= 0;
			v = (v << 1) | (u64x)(IsZero(coeff(m, 0)) ? 0 : 1);
		}
		return v;
	}
};

int main(int argc, char **argv) {
	const bool checkRounds = !(argc > 1 && std::string(argv[1]) == "--no-round-checks");
	const u64x F = ~0ULL;
	const u64x tv[5][4] = {{0, 0, 0, 0x818665aa0d02dfdaULL}, {F, 0, 0, 0x604ae6ca03c20adaULL}, {0, F, 0, 0x9fb51935fc3df524ULL},
	                       {0, 0, F, 0x78a54cbe737bb7efULL}, {0x0123456789abcdefULL, 0, 0xfedcba9876543210ULL, 0xae25ad3ca8fa9ccfULL}};
	for (auto &v : tv) if (plainPrince(v[0], v[1], v[2], NULL) != v[3]) { printf("plain PRINCE disagrees with a published test vector\n"); return 2; }
	const u64x pt = 0, key0 = F, key1 = 0;                  // the reference's run (Prince.cu:69-74)
	multiGPUs(1);
	Dhs dhs;
	dhs.setup(25, 2, 16, 25, 25, 21845);
	if (!getenv("CUHE_SCHED")) setScheduled(false);         // this client issues whole layers itself; CUHE_SCHED=1 in the environment runs it on scheduled gates all the same (tools/sched_soak.sh)
	printf("DHS(25,2,16,25,25,21845): n=%d nttLen=%d primes=%d evalKeys=%d\n", dhs.n, param.nttLen, param.numCrtPrime, param.numEvalKey);
	Machine M(dhs);
	std::vector<u64x> expect;
	plainPrince(pt, key0, key1, &expect);
	// encrypt and place: S = message bits, K = [k1 | k0]
	for (int i = 0; i < 192; ++i) {
		const int bit = i < 64 ? (int)((pt >> (63 - i)) & 1) : i < 128 ? (int)((key1 >> (127 - i)) & 1) : (int)((key0 >> (191 - i)) & 1);
		CuCtxt c; c.setLevel(0, 0, dhs.encryptBit(bit, 0)); c.x2c();
		uint32_t *dst = i < 64 ? M.S + (size_t)i * M.ct32(0) : M.K + (size_t)(i - 64) * M.ct32(0);
		CK(cuhe_hip_memcpy_d2d(0, dst, c.cRep(), M.ct32(0) * 4, NULL));
		CK(cuhe_hip_stream_sync(0, NULL));
	}
	std::vector<std::vector<int>> ident(64), mp = mPrimeSources(), mpSr(64), srInvMp(64);
	for (int i = 0; i < 64; ++i) ident[i] = {i};
	// M = SR o M': output nibble i of SR takes nibble SR[i] of M'(state); M^-1 = M' o SR^-1
	for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) mpSr[4 * i + k] = mp[4 * SR[i] + k];
	{
		int srInvSrc[64];                               // SR^-1: output bit 4*SR[i]+k comes from input bit 4*i+k
		for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) srInvSrc[4 * SR[i] + k] = 4 * i + k;
		for (int i = 0; i < 64; ++i) for (int s : mp[i]) srInvMp[i].push_back(srInvSrc[s]);
	}
	double paused = 0; int layer = 0;
	auto check = [&] {
		if (checkRounds) {
			CK(cuhe_hip_stream_sync(0, NULL));           // the layer's work is only enqueued: it belongs to the timed part
			const auto c0 = clk::now();
			bool constant; const u64x got = M.decryptState(constant);
			const bool ok = constant && got == expect[layer];
			printf("S-box layer %2d  level %2d  %016llx  %s\n", layer, M.level, got, ok ? "right" : "wrong");
			if (!ok) ++failures;
			paused += std::chrono::duration<double>(clk::now() - c0).count();
		}
		++layer;
	};
	// the tables of the linear layers (round constants, key additions, M', shift rows) are circuit constants: built once
	std::vector<std::vector<int>> k0terms(64), k0p(64);
	for (int i = 0; i < 64; ++i) { k0terms[i] = {64 + i}; k0p[i] = {64 + (i + 63) % 64}; }     // K[64..127] = k0
	k0p[63].push_back(64 + 0);                                                                   // k0' = (k0 >>> 1) ^ (k0 >> 63)
	const Csr whitenIn = Machine::linearTable(ident, RC[0], true, &k0terms);
	std::vector<Csr> fwdRound(6), keyRc(11);
	for (int i = 1; i <= 5; ++i) fwdRound[i] = Machine::linearTable(mpSr, RC[i], true, NULL);
	const Csr middle = Machine::linearTable(mp, 0, false, NULL), invMix = Machine::linearTable(srInvMp, 0, false, NULL);
	for (int i = 6; i <= 10; ++i) keyRc[i] = Machine::linearTable(ident, RC[i], true, NULL);
	const Csr whitenOut = Machine::linearTable(ident, RC[11], true, &k0p);
	CK(cuhe_hip_stream_sync(0, NULL));
	const auto t0 = clk::now();
	M.linear(whitenIn);                                  // state = m ^ k0 ^ k1 ^ RC0
	for (int i = 1; i <= 5; ++i) { M.sboxLayer(M.sboxFwd); check(); M.linear(fwdRound[i]); }
	M.sboxLayer(M.sboxFwd); check();
	M.linear(middle);
	M.sboxLayer(M.sboxInv); check();
	for (int i = 6; i <= 10; ++i) {
		M.linear(keyRc[i]);
		M.linear(invMix);
		M.sboxLayer(M.sboxInv); check();
	}
	M.linear(whitenOut);                                 // ^ RC11 ^ k1 ^ k0'
	CK(cuhe_hip_stream_sync(0, NULL));
	const double encSeconds = std::chrono::duration<double>(clk::now() - t0).count() - paused;
	bool constant; const u64x got = M.decryptState(constant);
	const u64x want = plainPrince(pt, key0, key1, NULL);
	const bool ok = constant && got == want && want == 0x9fb51935fc3df524ULL && M.level == 24;
	printf("homomorphic PRINCE: %016llx   expected %016llx   %s\n", got, want, ok ? "right" : "wrong");
	if (!ok) ++failures;
	printf("Prince Encryption: %.3f s on 1 GPU, gates on arrays of ciphertexts (round checks excluded)\n", encSeconds);
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
