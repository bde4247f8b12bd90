// test_cuhe_api.cpp -- exercises the C++ drop-in API (cuhe_amd/cxx/CuHE.h) the way the
// reference's DHS example does (examples/DHS/simple_DHS.cu:49-211, DHS.cu:212-252), against
// host ZZX arithmetic: t = a*b; t %= Phi_m; coefficients % q  (examples/DHS/DHS.cu:219-221).
// Runs on a GPU box; exit code 0 = all checks passed.
#include "CuHE.h"
#include "Relinearization.h"
#include "CuHEArray.h"
#include "cuhe_hip.h"
#include "Debug.h"
#include "DeviceManager.h"
#include "Operations.h"
#include <cstdio>
#include <vector>
using namespace cuHE;

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } else printf("ok: %s\n", what); } while (0)

static ZZX cyclotomic(int m) {
	auto mu = [](int n) { int r = 1; for (int p = 2; p * p <= n; ++p) if (n % p == 0) { n /= p; if (n % p == 0) return 0; r = -r; } if (n > 1) r = -r; return r; };
	std::vector<long long> a(2 * m + 2, 0); int len = 1; a[0] = 1;
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == 1) { for (int i = len - 1; i >= 0; --i) { a[i + d] += a[i]; a[i] = -a[i]; } len += d; }
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == -1) { for (int i = 0; i < len - d; ++i) a[i] = (i >= d ? a[i - d] : 0) - a[i]; len -= d; }
	ZZX r; for (int i = 0; i < len; ++i) SetCoeff(r, i, to_ZZ((long)a[i])); return r;
}
static ZZX randomPoly(int n, const ZZ &q) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, RandomBnd(q)); return r; }
static ZZX reduceCoeffs(const ZZX &a, const ZZ &q, int n) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, coeff(a, i) % q); return r; }
static ZZX hostMul(const ZZX &a, const ZZX &b, const ZZX &phi, const ZZ &q, int n) { ZZX t = a * b; t %= phi; return reduceCoeffs(t, q, n); }

int main() {
	SetSeed(to_ZZ(20260926));
	const int d = 3, p = 2, w = 8, mn = 40, cut = 20, m = 1155;
	setParameters(d, p, w, mn, cut, m);
	CHECK(param.modLen == 480 && param.nttLen == 16384 && param.numCrtPrime == 4 && param.numEvalKey == 10, "setParameters derives the reference's values");
	ZZX phi = cyclotomic(m);
	CHECK(deg(phi) == param.modLen, "cyclotomic degree");
	std::vector<ZZ> q(d);
	// a client that composes the pre-computation itself, in the order of the reference's initCuHE (cuhe/CuHE.cu:47-49), with a
	// modulus that is NOT the cyclotomic initCrt has to assume: x^480 + 1 here, then the real one through initCuHE below
	{
		ZZX other; SetCoeff(other, param.modLen, 1); SetCoeff(other, 0, 1);
		std::vector<ZZ> q2(d);
		initNtt();
		initCrt(q2.data());
		initBarrett(other);
		const int n2 = param.modLen;
		ZZX a = randomPoly(n2, q2[0]), b = randomPoly(n2, q2[0]), c;
		mulZZX(c, a, b, 0, 0);
		CHECK(c == hostMul(a, b, other, q2[0], n2), "initNtt ; initCrt ; initBarrett composed by the client: mulZZX modulo the modulus given to initBarrett");
		genCrtPrimes(); genIcrt(); loadIcrtConst(0, 0);          // no-ops on an initialised library
		initCuHE(q.data(), phi);
		CHECK(q2[0] == q[0] && q2[d - 1] == q[d - 1], "initCrt alone returns the coefficient moduli initCuHE returns");
	}
	const int n = param.modLen;
	bool chain = true;
	for (int i = 0; i < d; ++i) chain = chain && NumBits(q[i]) <= param._logCoeff(i) && (i == 0 || (q[i - 1] % q[i] == to_ZZ(0) && q[i] < q[i - 1]));
	CHECK(chain, "initCuHE returns a decreasing divisor chain of coefficient moduli");

	// ---- mulZZX (DHS.cu:218) at two levels
	for (int lvl : {0, 2}) {
		ZZX a = randomPoly(n, q[lvl]), b = randomPoly(n, q[lvl]), c;
		mulZZX(c, a, b, lvl, 0);
		CHECK(c == hostMul(a, b, phi, q[lvl], n), lvl == 0 ? "mulZZX level 0" : "mulZZX level 2");
	}
	// ---- mulZZXBatch (addition): five independent products in one call, level 1
	{
		std::vector<ZZX> a(5), b(5), c(5);
		bool ok = true;
		for (int i = 0; i < 5; ++i) { a[i] = randomPoly(n, q[1]); b[i] = randomPoly(n, q[1]); }
		mulZZXBatch(c.data(), a.data(), b.data(), 5, 1, 0);
		for (int i = 0; i < 5; ++i) ok = ok && c[i] == hostMul(a[i], b[i], phi, q[1], n);
		CHECK(ok, "mulZZXBatch equals five host products");
	}
	// ---- domain conversions round trip + cXor in CRT and NTT domains (simple_DHS.cu checkXor)
	{
		ZZX a = randomPoly(n, q[0]), b = randomPoly(n, q[0]);
		CuCtxt ca, cb, cz;
		ca.setLevel(0, 0, a); cb.setLevel(0, 0, b);
		ca.x2c(); cb.x2c();
		cXor(cz, ca, cb);
		cz.x2z();
		CHECK(cz.zRep() == reduceCoeffs(a + b, q[0], n), "cXor in the CRT domain");
		ca.x2n(); cb.x2n();
		CuCtxt cy;
		cXor(cy, ca, cb);
		cy.x2z();
		CHECK(cy.zRep() == reduceCoeffs(a + b, q[0], n), "cXor in the NTT domain");
		ca.x2z();
		CHECK(ca.zRep() == a, "ZZX -> RAW -> CRT -> NTT -> CRT -> RAW -> ZZX round trip");
	}
	// ---- the CRT rows a polynomial was transformed from are kept while its NTT rows are only read (round 5): the way back must give
	//      the same rows, and every write to the NTT rows must end the shortcut
	{
		ZZX a = randomPoly(n, q[0]), b = randomPoly(n, q[0]);
		CuCtxt ca, cb;
		ca.setLevel(0, 0, a); cb.setLevel(0, 0, b);
		ca.x2n(); cb.x2n();
		CuCtxt prod; cAnd(prod, ca, cb);                       // ca, cb only read
		ca.x2c(); CHECK(ca.cRep() != NULL && ca.domain() == 2, "x2n ; (read) ; x2c comes back to the CRT domain");
		ca.x2z(); CHECK(ca.zRep() == a, "x2n ; (read) ; x2c returns the rows it started from");
		cb.x2c(); cb.x2n();                                    // back and forth again
		cXor(cb, cb, cb);                                      // written in place in the NTT domain: 2 b
		cb.x2z(); CHECK(cb.zRep() == reduceCoeffs(b + b, q[0], n), "a sum written into NTT rows is not shadowed by the rows kept from before");
		CuCtxt cc; cc.setLevel(0, 0, a); cc.x2n();
		CHECK(cc.cRep() == NULL, "cRep() is NULL in the NTT domain, as in the reference");
		uint64 *raw = cc.nRep();                               // a raw pointer was handed out: the library must assume it is written through
		CHECK(raw != NULL, "nRep() in the NTT domain");
		cAnd(cc, cc, cc);                                      // in place: a * a
		cc.x2z(); CHECK(cc.zRep() == hostMul(a, a, phi, q[0], n), "a product formed in place is what x2z returns");
		CuCtxt cd; cd.setLevel(0, 0, a); cd.x2n();
		CuCtxt ce; cAnd(ce, cd, cd);                           // cd only read
		cd.modSwitch();                                        // from the NTT domain, through the kept rows
		CuCtxt cf; cf.setLevel(0, 0, a); cf.x2c(); cf.modSwitch();
		cd.x2z(); cf.x2z();
		CHECK(cd.zRep() == cf.zRep() && cd.level() == 1, "modSwitch from the NTT domain equals modSwitch from the CRT domain");
	}
	// ---- cNot (simple_DHS.cu:98) : adds modMsg-1 to the constant term
	{
		ZZX a = randomPoly(n, q[0]);
		CuCtxt ca; ca.setLevel(0, 0, a); ca.x2c();
		cNot(ca, ca);
		ca.x2z();
		ZZX want = a; SetCoeff(want, 0, (coeff(a, 0) + to_ZZ(param.modMsg - 1)) % q[0]);
		CHECK(ca.zRep() == want, "cNot");
	}
	// ---- cAnd with a plaintext polynomial (CuPtxt, nttMulNX1)
	{
		ZZX a = randomPoly(n, q[0]), pt;
		for (int i = n - 1; i >= 0; --i) SetCoeff(pt, i, RandomBnd(to_ZZ(2)));
		CuCtxt ca; ca.setLevel(0, 0, a); ca.x2n();
		CuPtxt cp; cp.setLogq(param.logMsg, 0, pt); cp.x2n();
		CuCtxt cz; cAnd(cz, ca, cp);
		cz.x2z();
		CHECK(cz.zRep() == hostMul(a, pt, phi, q[0], n), "cAnd(ciphertext, plaintext)");
	}
	// ---- cAnd + relin + modSwitch (simple_DHS.cu checkAnd: cAnd; relin; modSwitch)
	{
		std::vector<ZZX> ek(param.numEvalKey);
		for (auto &e : ek) e = randomPoly(n, q[0]);
		initRelinearization(ek.data());
		ZZX a = randomPoly(n, q[0]), b = randomPoly(n, q[0]);
		CuCtxt ca, cb, cz;
		ca.setLevel(0, 0, a); cb.setLevel(0, 0, b);
		ca.x2n(); cb.x2n();
		cAnd(cz, ca, cb);
		cz.relin();
		CHECK(cz.domain() == 2 && cz.level() == 0, "relin leaves a CRT-domain ciphertext at the same level");
		// expected: sum_j window_j(a*b) * ek_j  mod Phi, mod q0
		ZZX prod = hostMul(a, b, phi, q[0], n), acc;
		const ZZ base = power2_ZZ(param.logRelin);
		for (int j = 0; j < param._numEvalKey(0); ++j) {
			ZZX win; ZZ sh = power(base, j);
			for (int i = n - 1; i >= 0; --i) SetCoeff(win, i, (coeff(prod, i) / sh) % base);
			acc += win * ek[j];
		}
		acc %= phi;
		ZZX want = reduceCoeffs(acc, q[0], n);
		CuCtxt keep; copy(keep, cz);
		cz.x2z();
		CHECK(cz.zRep() == want, "cAnd + relin equals the windowed key-switch sum");
		// binary key cache: save, clobber the resident keys with other ones, load, relinearise again
		{
			const char *path = "/tmp/cuhe_evalkeys.bin";
			saveRelinearization(path);
			std::vector<ZZX> other(param.numEvalKey);
			for (auto &e : other) e = randomPoly(n, q[0]);
			initRelinearization(other.data());
			CuCtxt c2; cAnd(c2, ca, cb); c2.relin(); c2.x2z();
			const bool differs = !(c2.zRep() == want);
			const bool loaded = loadRelinearization(path);
			CuCtxt c3; cAnd(c3, ca, cb); c3.relin(); c3.x2z();
			CHECK(differs && loaded && c3.zRep() == want, "saveRelinearization / loadRelinearization restore the NTT-domain keys");
			remove(path);
			CHECK(!loadRelinearization(path), "loading a missing key cache reports failure");
		}
		// modSwitch on the relinearised ciphertext
		keep.modSwitch();
		CHECK(keep.level() == 1 && keep.logq() == param._logCoeff(1), "modSwitch advances the level");
		keep.x2z();
		const ZZ pt = q[0] / q[1];
		ZZX ms;
		for (int i = n - 1; i >= 0; --i) {
			ZZ v = coeff(want, i), dl = v % pt;
			if (dl % to_ZZ(2) != to_ZZ(0)) dl = (dl > (pt - to_ZZ(1)) / to_ZZ(2)) ? dl - pt : dl + pt;
			SetCoeff(ms, i, ((v - dl) / pt) % q[1]);
		}
		CHECK(keep.zRep() == ms, "modSwitch = (c - delta)/p_t with delta = c mod p_t made even");
	}
	// ---- gates on arrays of ciphertexts (CuHEArray.h, addition) against the single-ciphertext gates
	{
		const int C = 5;
		std::vector<ZZX> z(C), w(C);
		CuCtxtArray arr, other;
		arr.create(C, 0, 2); other.create(C, 0, 2);
		for (int i = 0; i < C; ++i) {
			z[i] = randomPoly(n, q[0]); w[i] = randomPoly(n, q[0]);
			CuCtxt c; c.setLevel(0, 0, z[i]); c.x2c(); arr.put(i, c);
			CuCtxt d; d.setLevel(0, 0, w[i]); d.x2c(); other.put(i, d);
		}
		// cXor / cNot over index lists: out[0] = z0 + z2 + w1 + 1, out[1] = z1 + z4
		CuIndexTable off, list, one;
		off.set({0, 3, 5}); list.set({0, 2, C + 1, 1, 4}); one.set({1, 0});
		CuCtxtArray sums;
		cXor(sums, arr, &other, off, list, one);
		CuCtxt s0, s1; sums.get(s0, 0); sums.get(s1, 1); s0.x2z(); s1.x2z();
		ZZX want0 = z[0] + z[2] + w[1]; SetCoeff(want0, 0, coeff(want0, 0) + 1);
		CHECK(s0.zRep() == reduceCoeffs(want0, q[0], n) && s1.zRep() == reduceCoeffs(z[1] + z[4], q[0], n), "arrays: cXor / cNot over index lists");
		// cAnd over index pairs, relin, modSwitch: four chains in three calls
		arr.x2n();
		CuIndexTable ia, ib; ia.set({0, 1, 2, 3}); ib.set({1, 2, 3, 4});
		CuCtxtArray prod;
		cAnd(prod, arr, ia, ib);
		prod.relin();
		prod.modSwitch();
		bool same = prod.level() == 1 && prod.domain() == 2 && prod.count() == 4;
		for (int t = 0; t < 4 && same; ++t) {
			CuCtxt a, b, p;
			a.setLevel(0, 0, z[t]); b.setLevel(0, 0, z[t + 1]);
			a.x2n(); b.x2n();
			cAnd(p, a, b); p.relin(); p.modSwitch(); p.x2z();
			CuCtxt g; prod.get(g, t); g.x2z();
			same = same && g.level() == 1 && g.zRep() == p.zRep();
		}
		CHECK(same, "arrays: cAnd over index pairs + relin + modSwitch = four single-ciphertext chains");
		// plain conversion round trip of the whole array
		arr.x2c();
		CuCtxt back; arr.get(back, 3); back.x2z();
		CHECK(back.zRep() == z[3], "arrays: CRT -> NTT -> CRT round trip");
	}
	// ---- the second tier, cuhe/Operations.h:42-108: raw-pointer drivers called directly, the way cuhe/CuHE.cu strings
	//      them together (crt / ntt / nttMul / inttMod / icrt = the body of mulZZX, CuHE.cu:259-268 + :350-408), and the
	//      other two roads to a reduced product: inttDoubleDeg + barrett(dst, src) and inttHold + barrett(dst)
	{
		selectDevice(0);
		ZZX a = randomPoly(n, q[0]), b = randomPoly(n, q[0]);
		const ZZX want = hostMul(a, b, phi, q[0], n);
		CuCtxt ca, cb; ca.setLevel(0, 0, a); cb.setLevel(0, 0, b);
		ca.x2r(); cb.x2r();
		const int logq = ca.logq(), np = param._numCrtPrime(0);
		const size_t crtBytes = (size_t)np * param.crtLen * sizeof(uint32), nttBytes = (size_t)np * param.nttLen * sizeof(uint64);
		uint32 *ac = (uint32 *)deviceMalloc(crtBytes), *bc = (uint32 *)deviceMalloc(crtBytes), *rc = (uint32 *)deviceMalloc(crtBytes);
		uint32 *dbl = (uint32 *)deviceMalloc((size_t)np * param.nttLen * sizeof(uint32));
		uint64 *an = (uint64 *)deviceMalloc(nttBytes), *bn = (uint64 *)deviceMalloc(nttBytes), *pn = (uint64 *)deviceMalloc(nttBytes);
		// crt() writes the modLen coefficients of the ring and ntt() reads whole rows of crtLen: the rest of a CRT row has to BE zero, which is why the
		// reference creates its representations with cudaMalloc + cudaMemset (cuhe/CuHE.cu, *RepCreate) -- and a block that comes back from the pool holds
		// whatever its last owner left there (with scheduled gates that made this section fail in three runs out of four until the rows were cleared)
		CSC(cuhe_hip_memset_async(0, ac, 0, crtBytes, 0)); CSC(cuhe_hip_memset_async(0, bc, 0, crtBytes, 0)); CSC(cuhe_hip_memset_async(0, rc, 0, crtBytes, 0));
		CSC(cuhe_hip_memset_async(0, dbl, 0, (size_t)np * param.nttLen * sizeof(uint32), 0));
		crt(ac, ca.rRep(), logq, 0); crt(bc, cb.rRep(), logq, 0);
		ntt(an, ac, logq, 0); ntt(bn, bc, logq, 0);
		nttMul(pn, an, bn, logq, 0);
		auto rawToZZX = [&](uint32 *crtRows) { CuCtxt o; o.setLevel(0, 1, 0); icrt(o.rRep(), crtRows, logq, 0); o.x2z(); return o.zRep(); };
		inttMod(rc, pn, logq, 0);
		CHECK(rawToZZX(rc) == want, "Operations.h: crt, ntt, nttMul, inttMod, icrt by hand = mulZZX");
		inttDoubleDeg(dbl, pn, logq, 0);
		barrett(rc, dbl, 0, 0);
		CHECK(rawToZZX(rc) == want, "Operations.h: inttDoubleDeg + barrett(dst, src)");
		inttHold(pn, logq, 0);
		barrett(rc, 0, 0);
		CHECK(rawToZZX(rc) == want, "Operations.h: inttHold + barrett(dst)");
		// sums in both domains, and a product with one polynomial for all primes' rows replaced by a full set (nttAdd)
		crtAdd(rc, ac, bc, logq, 0);
		CHECK(rawToZZX(rc) == reduceCoeffs(a + b, q[0], n), "Operations.h: crtAdd");
		nttAdd(pn, an, bn, logq, 0);
		intt(rc, pn, logq, 0);
		CHECK(rawToZZX(rc) == reduceCoeffs(a + b, q[0], n), "Operations.h: nttAdd + intt");
		crtAddInt(bc, bc, 5, logq, 0);              // writes the constant terms only (cuhe/Base.cu:1096-1100): used in place
		{ ZZX w5 = b; SetCoeff(w5, 0, (coeff(b, 0) + 5) % q[0]); CHECK(rawToZZX(bc) == w5, "Operations.h: crtAddInt (in place)"); }
		// single-row forms: _intt(_ntt(row)) gives the row back, zero-padded to nttLen
		{
			std::vector<uint32> row(param.crtLen), back(param.nttLen);
			_ntt(an, ac, 0);
			_intt(dbl, an, 0, 0);
			CSC(cuhe_hip_memcpy_d2h(0, row.data(), ac, row.size() * 4, 0));
			CSC(cuhe_hip_memcpy_d2h(0, back.data(), dbl, back.size() * 4, 0));
			CSC(cuhe_hip_stream_sync(0, 0));
			bool same = true;
			for (int i = 0; i < param.nttLen; ++i) same = same && back[i] == (i < param.crtLen ? row[i] : 0u);
			CHECK(same, "Operations.h: _intt(_ntt(row)) = row");
		}
		for (void *ptr : {(void *)ac, (void *)bc, (void *)rc, (void *)dbl, (void *)an, (void *)bn, (void *)pn}) deviceFree(ptr);
	}
	// ---- ownership rules the examples rely on (Prince.cu:298-319: explicit destructor, then scope exit)
	{
		startAllocator();
		ZZX a = randomPoly(n, q[0]);
		CuCtxt ca; ca.setLevel(0, 0, a); ca.x2n();
		CuCtxt cb; copy(cb, ca);
		moveTo(cb, 0);
		ca.~CuCtxt();
		cb.x2z();
		CHECK(cb.zRep() == a, "copy / moveTo(same device) / explicit destructor with the pooled allocator");
		stopAllocator();
	}
	// ---- SUMS of products in the NTT domain on a ring whose negacyclic representation has room for TWO products
	// (x^16384 + 1, 24-bit primes: 2 n p^2 just below P / 2; ONE on x^32768 + 1).  Operands with every residue p - 1 drive the
	// integer coefficients of each product to n (p - 1)^2; cXor has to notice when a sum leaves the range the inverse
	// transform recovers and reduce the operands first.
	{
		cuhe_hip_shutdown();
		setParameters(3, 2, 16, 48, 24, 32768);
		const int n2 = param.modLen;
		ZZX xn1; SetCoeff(xn1, n2, 1); SetCoeff(xn1, 0, 1);
		std::vector<ZZ> q2(3);
		initCuHE(q2.data(), xn1);
		CHECK(cuhe_hip_ct_negacyclic() == 1 && cuhe_hip_ct_prod_headroom() == 2, "x^16384 + 1 with 24-bit primes: negacyclic, room for two products");
		ZZX a, b;
		for (int i = 0; i < n2; ++i) { SetCoeff(a, i, q2[0] - 1); SetCoeff(b, i, (i & 1) ? q2[0] - 1 : to_ZZ(1)); }
		CuCtxt ca, cb, p1, p2, sum;
		ca.setLevel(0, 0, a); cb.setLevel(0, 0, b);
		ca.x2n(); cb.x2n();
		cAnd(p1, ca, ca);
		cAnd(p2, ca, cb);
		cXor(sum, p1, p2);
		CHECK(sum.domain() == 3, "cXor of two products answers in the NTT domain");
		sum.x2z();
		const ZZX aaHost = hostMul(a, a, xn1, q2[0], n2);        // (the schoolbook product of 16384 coefficients on the fallback big integer takes ~17 s: once)
		ZZX want = reduceCoeffs(aaHost + hostMul(a, b, xn1, q2[0], n2), q2[0], n2);
		CHECK(sum.zRep() == want, "a*a + a*b added in the NTT domain (two products: inside the headroom)");
		{
			CuCtxt q1, q2c, q3, s12, s123;
			cAnd(q1, ca, ca); cAnd(q2c, ca, ca); cAnd(q3, ca, ca);
			cXor(s12, q1, q2c);
			cXor(s123, s12, q3);                                 // three products: beyond the headroom, the operands are reduced first
			CHECK(s123.domain() == 3, "cXor beyond the headroom still answers in the NTT domain");
			s123.x2z();
			const ZZX &aa = aaHost;
			CHECK(s123.zRep() == reduceCoeffs(aa + aa + aa, q2[0], n2), "a*a + a*a + a*a added in the NTT domain (three products: beyond the headroom)");
		}
		{
			//  the count of summed products survives a sum with a PLAINTEXT and a trip through an array: p1 + p2
			// (two products), + plaintext (still two), + a third product has to take the reducing path
			ZZX one; SetCoeff(one, 0, 1);
			CuPtxt pt; pt.setLogq(param.logMsg, 0, one); pt.x2n();
			CuCtxt q1, q2c, q3, s12, t, u;
			cAnd(q1, ca, ca); cAnd(q2c, ca, ca); cAnd(q3, ca, ca);
			cXor(s12, q1, q2c);
			cXor(t, s12, pt);
			CHECK(t.isProd() && t.prodTerms() == 2, "a sum of two products plus a plaintext still counts two products");
			cXor(u, t, q3);
			u.x2z();
			const ZZX &aa = aaHost;
			CHECK(u.zRep() == reduceCoeffs(aa + aa + aa + one, q2[0], n2), "(a*a + a*a + 1) + a*a: exact although the plaintext sum sat in between");
			CuCtxt v, w, q4;
			cXor(v, q1, q2c);
			cXor(v, v, pt);                                      // in place
			CHECK(v.prodTerms() == 2, "in-place sum with a plaintext keeps the count");
			CuCtxtArray arr; arr.create(1, 0, 3, 0);
			arr.put(0, v);
			arr.get(w, 0);
			CHECK(w.isProd() && w.prodTerms() == 2, "the count survives CuCtxtArray put / get");
			cAnd(q4, ca, ca);
			cXor(w, w, q4);
			w.x2z();
			CHECK(w.zRep() == reduceCoeffs(aa + aa + aa + one, q2[0], n2), "array round trip, then a third product: exact");
		}
		{
			// the result may be the SECOND operand (the reference resets it before reading it: cuhe/CuHE.cu:108-111,141-145)
			ZZX ra2 = randomPoly(n2, q2[0]), rb2 = randomPoly(n2, q2[0]);
			CuCtxt x, y;
			x.setLevel(0, 0, ra2); y.setLevel(0, 0, rb2); x.x2n(); y.x2n();
			cXor(y, x, y);
			CuCtxt t; copy(t, y); t.x2z();
			CHECK(t.zRep() == reduceCoeffs(ra2 + rb2, q2[0], n2), "cXor(out = second operand) in the NTT domain");
			y.x2c(); x.x2c();
			cXor(y, x, y);
			copy(t, y); t.x2z();
			CHECK(t.zRep() == reduceCoeffs(ra2 + ra2 + rb2, q2[0], n2), "cXor(out = second operand) in the CRT domain");
			x.x2n(); y.x2n();
			cAnd(y, x, y);
			y.x2z();
			CHECK(y.zRep() == hostMul(ra2, reduceCoeffs(ra2 + ra2 + rb2, q2[0], n2), xn1, q2[0], n2), "cAnd(out = second operand)");
		}
		ZZX ra = randomPoly(n2, q2[0]), rb = randomPoly(n2, q2[0]);
		CuCtxt cra, crb, s2;
		cra.setLevel(0, 0, ra); crb.setLevel(0, 0, rb); cra.x2n(); crb.x2n();
		cXor(s2, cra, crb);                                      // no products involved: plain NTT-domain addition
		s2.x2z();
		CHECK(s2.zRep() == reduceCoeffs(ra + rb, q2[0], n2), "cXor of two fresh NTT-domain ciphertexts");
	}
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
