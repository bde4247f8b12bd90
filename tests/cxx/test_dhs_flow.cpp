// test_dhs_flow.cpp -- a complete DHS/LTV homomorphic-encryption flow through the C++ drop-in API
// (cuhe_amd/cxx/CuHE.h): key generation, encryption, homomorphic XOR / NOT / AND with relinearisation and
// modulus switching, decryption.  It plays the role of the reference's examples/DHS/simple_DHS.cu
// (checkXor :49, checkNot :90, checkAnd :130) with a client written from scratch on top of this
// repository's API (the reference's CuDHS class needs NTL's ZZ_pE/ZZ_pX machinery, which is not in the
// image).  A wrong relinearisation, modulus switch, reduction mod Phi_m or CRT/ICRT makes decryption fail,
// so "right" here is an end-to-end semantic check of the whole hot path, independent of the oracle.
//
// Scheme (examples/DHS/DHS.cu:212-372, restated):  f = 2f'+1 invertible in Z_q0[x]/Phi_m, pk = 2 g f^-1,
// Enc(m) = pk s + 2e + m, Dec(c) = centred(f c) mod 2, ek_j = pk s_j + 2 e_j + f 2^(w j),
// AND = product, relin = sum_j window_j(c) ek_j, then modSwitch.  No batching: a message is a polynomial
// with binary coefficients, XOR/AND act on it as addition/multiplication in Z_2[x]/Phi_m.
//
// usage: test_dhs_flow [d p w min cut m]   (default: the reference example's (5,2,1,61,20,8191))
#include "CuHE.h"
#include "cuhe_hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace cuHE;
typedef long long i64;
typedef unsigned long long u64x;

static int failures = 0;
static void report(const char *what, bool ok) { printf("%s\t%s\n", what, ok ? "right" : "wrong"); if (!ok) ++failures; }

static std::vector<i64> cyclotomicInts(int m) {
	auto mu = [](int n) { int r = 1; for (int p = 2; p * p <= n; ++p) if (n % p == 0) { n /= p; if (n % p == 0) return 0; r = -r; } if (n > 1) r = -r; return r; };
	std::vector<i64> a(2 * m + 2, 0); int len = 1; a[0] = 1;
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == 1) { for (int i = len - 1; i >= 0; --i) { a[i + d] += a[i]; a[i] = -a[i]; } len += d; }
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == -1) { for (int i = 0; i < len - d; ++i) a[i] = (i >= d ? a[i - d] : 0) - a[i]; len -= d; }
	a.resize(len); return a;
}

// ---------------------------------------------------------------- F_p[x] helpers for the key inverse
static u64x powmod(u64x b, u64x e, u64x p) { u64x r = 1; b %= p; while (e) { if (e & 1) r = r * b % p; b = b * b % p; e >>= 1; } return r; }
// inverse of f modulo (phi, p) by the extended Euclidean algorithm; false if gcd(f, phi) != 1 over F_p
static bool invertModPrime(std::vector<unsigned> &inv, const std::vector<unsigned> &f, const std::vector<unsigned> &phi, unsigned p) {
	typedef std::vector<u64x> Poly;
	auto trim = [](Poly &a) { while (!a.empty() && a.back() == 0) a.pop_back(); };
	Poly r0(phi.begin(), phi.end()), r1(f.begin(), f.end()), t0, t1(1, 1);
	trim(r0); trim(r1);
	while (!r1.empty()) {
		// r0 = q r1 + r2, t2 = t0 - q t1, one quotient term at a time
		const u64x lead = powmod(r1.back(), p - 2, p);
		while (r0.size() >= r1.size()) {
			const size_t sh = r0.size() - r1.size();
			const u64x c = r0.back() * lead % p;
			for (size_t i = 0; i < r1.size(); ++i) r0[i + sh] = (r0[i + sh] + (p - c) * r1[i]) % p;
			if (t0.size() < t1.size() + sh) t0.resize(t1.size() + sh, 0);
			for (size_t i = 0; i < t1.size(); ++i) t0[i + sh] = (t0[i + sh] + (p - c) * t1[i]) % p;
			trim(r0);
			if (r0.empty()) break;
		}
		std::swap(r0, r1); std::swap(t0, t1);
	}
	if (r0.size() != 1) return false;                       // gcd has positive degree
	const u64x g = powmod(r0[0], p - 2, p);
	// t0 may have degree >= deg(phi) only transiently; reduce modulo phi (monic)
	Poly t = t0; const size_t n = phi.size() - 1;
	for (size_t k = t.size(); k-- > n;) { const u64x c = t[k]; if (!c) continue; for (size_t i = 0; i <= n; ++i) t[k - n + i] = (t[k - n + i] + (p - c) * phi[i]) % p; }
	inv.assign(n, 0);
	for (size_t i = 0; i < n && i < t.size(); ++i) inv[i] = (unsigned)(t[i] * g % p);
	return true;
}

// ---------------------------------------------------------------- the scheme
struct Dhs {
	int n, depth, np;
	std::vector<ZZ> q;                 // coefficient modulus per level
	std::vector<unsigned> primes;
	std::vector<i64> phi;
	ZZX phiZ;
	std::vector<ZZX> pk, sk, ek;

	ZZX sample() { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, RandomBnd(to_ZZ(3)) - to_ZZ(1)); return r; }
	ZZX reduce(const ZZX &a, const ZZ &m) { ZZX r; for (long i = deg(a); i >= 0; --i) SetCoeff(r, i, coeff(a, i) % m); return r; }
	ZZX scaleAdd(const ZZX &a, long s, const ZZX &b) { ZZX r; long d = std::max(deg(a), deg(b)); for (long i = d; i >= 0; --i) SetCoeff(r, i, coeff(a, i) * to_ZZ(s) + coeff(b, i)); return r; }

	bool invert(ZZX &finv, const ZZX &f) {
		// per CRT prime (Z_q0[x]/Phi is the product of the F_p[x]/Phi), then lift the coefficients
		std::vector<std::vector<unsigned>> rows(np);
		for (int i = 0; i < np; ++i) {
			const unsigned p = primes[i];
			std::vector<unsigned> fp(n), php(n + 1);
			for (int k = 0; k < n; ++k) fp[k] = (unsigned)to_long(coeff(f, k) % to_ZZ((long)p));
			for (int k = 0; k <= n; ++k) php[k] = (unsigned)(((phi[k] % (i64)p) + p) % p);
			if (!invertModPrime(rows[i], fp, php, p)) return false;
		}
		std::vector<ZZ> lift(np);
		for (int i = 0; i < np; ++i) {
			const ZZ mi = q[0] / to_ZZ((long)primes[i]);
			const u64x bi = powmod((u64x)to_long(mi % to_ZZ((long)primes[i])), primes[i] - 2, primes[i]);
			lift[i] = mi * to_ZZ((long)bi);
		}
		clear(finv);
		for (int k = n - 1; k >= 0; --k) {
			ZZ v;
			for (int i = 0; i < np; ++i) v += lift[i] * to_ZZ((long)rows[i][k]);
			SetCoeff(finv, k, v % q[0]);
		}
		return true;
	}

	void setup(int d, int p, int w, int mn, int cut, int m) {
		setParameters(d, p, w, mn, cut, m);
		n = param.modLen; depth = param.depth; np = param.numCrtPrime;
		phi = cyclotomicInts(m);
		for (size_t i = 0; i < phi.size(); ++i) if (phi[i]) SetCoeff(phiZ, (long)i, to_ZZ((long)phi[i]));
		q.resize(depth);
		initCuHE(q.data(), phiZ);
		primes.resize(np);
		if (cuhe_hip_get_crt_primes(primes.data(), np) != 0) { printf("cannot read the CRT primes\n"); exit(2); }
		// keys (DHS.cu:286-322)
		ZZX f, finv, g;
		for (;;) {
			f = scaleAdd(sample(), param.modMsg, ZZX());
			SetCoeff(f, 0, coeff(f, 0) + to_ZZ(1));
			f = reduce(f, q[0]);
			if (invert(finv, f)) break;
		}
		g = reduce(sample(), q[0]);
		pk.resize(depth); sk.resize(depth);
		sk[0] = f;
		mulZZX(pk[0], g, finv, 0, 0, 0);
		pk[0] = reduce(scaleAdd(pk[0], param.modMsg, ZZX()), q[0]);
		for (int i = 1; i < depth; ++i) { sk[i] = reduce(sk[i - 1], q[i]); pk[i] = reduce(pk[i - 1], q[i]); }
		// evaluation keys (DHS.cu:323-345)
		ek.resize(param.numEvalKey);
		ZZ tw = to_ZZ(1); const ZZ wbase = power2_ZZ(param.logRelin);
		for (int j = 0; j < param.numEvalKey; ++j) {
			ZZX tp; for (int k = n - 1; k >= 0; --k) SetCoeff(tp, k, (coeff(sk[0], k) * tw) % q[0]);
			ZZX s = reduce(sample(), q[0]), e = sample(), t;
			mulZZX(t, pk[0], s, 0, 0, 0);
			ek[j] = reduce(scaleAdd(e, param.modMsg, t) + tp, q[0]);
			tw *= wbase;
		}
		initRelinearization(ek.data());
	}
	ZZX encrypt(const ZZX &msg, int lvl) {
		ZZX s = reduce(sample(), q[lvl]), e = sample(), t;
		mulZZX(t, pk[lvl], s, lvl, 0, 0);
		return reduce(scaleAdd(e, param.modMsg, t) + msg, q[lvl]);
	}
	ZZX decrypt(const ZZX &c, int lvl) {
		ZZX t, out;
		mulZZX(t, reduce(c, q[lvl]), sk[lvl], lvl, 0, 0);
		const ZZ half = (q[lvl] - to_ZZ(1)) / to_ZZ(2);
		for (long i = deg(t); i >= 0; --i) {
			ZZ x = coeff(t, i);
			if (x > half) x -= q[lvl];
			SetCoeff(out, i, x % to_ZZ(param.modMsg));
		}
		return out;
	}
};

static ZZX randomBits(int n) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, RandomBnd(to_ZZ(2))); return r; }
// (a * b mod Phi) mod 2 with machine integers
static ZZX mulMod2(const ZZX &a, const ZZX &b, const std::vector<i64> &phi) {
	const int n = (int)phi.size() - 1;
	std::vector<i64> t(2 * n, 0);
	for (int i = 0; i < n; ++i) if (!IsZero(coeff(a, i))) for (int j = 0; j < n; ++j) if (!IsZero(coeff(b, j))) ++t[i + j];
	for (int k = 2 * n - 1; k >= n; --k) { const i64 c = t[k]; if (!c) continue; for (int i = 0; i <= n; ++i) t[k - n + i] -= c * phi[i]; }
	ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, to_ZZ((long)(((t[i] % 2) + 2) % 2)));
	return r;
}
static ZZX addMod2(const ZZX &a, const ZZX &b, int n) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, (coeff(a, i) + coeff(b, i)) % to_ZZ(2)); return r; }

int main(int argc, char **argv) {
	int prm[6] = {5, 2, 1, 61, 20, 8191};
	if (argc == 7) for (int i = 0; i < 6; ++i) prm[i] = atoi(argv[i + 1]);
	SetSeed(to_ZZ(987654321));
	multiGPUs(1);
	Dhs dhs;
	dhs.setup(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
	const int n = dhs.n;
	printf("DHS(%d,%d,%d,%d,%d,%d): n=%d nttLen=%d primes=%d evalKeys=%d\n", prm[0], prm[1], prm[2], prm[3], prm[4], prm[5], n, param.nttLen, param.numCrtPrime, param.numEvalKey);

	ZZX x0 = randomBits(n), x1 = randomBits(n), x2 = randomBits(n);
	ZZX y0 = dhs.encrypt(x0, 0), y1 = dhs.encrypt(x1, 0), y2 = dhs.encrypt(x2, 0);
	report("dec(enc)", dhs.decrypt(y0, 0) == x0 && dhs.decrypt(y1, 0) == x1);

	{	// checkXor: NTT domain and CRT domain
		CuCtxt a, b, z; a.setLevel(0, 0, y0); b.setLevel(0, 0, y1);
		a.x2n(); b.x2n(); cXor(z, a, b); z.x2z();
		bool ok = dhs.decrypt(z.zRep(), 0) == addMod2(x0, x1, n);
		CuCtxt c, d, u; c.setLevel(0, 0, y0); d.setLevel(0, 0, y1);
		c.x2c(); d.x2c(); cXor(u, c, d); u.x2z();
		report("xor", ok && dhs.decrypt(u.zRep(), 0) == addMod2(x0, x1, n));
	}
	{	// checkNot
		CuCtxt a; a.setLevel(0, 0, y0); a.x2c(); cNot(a, a); a.x2z();
		ZZX want = x0; SetCoeff(want, 0, (coeff(x0, 0) + to_ZZ(1)) % to_ZZ(2));
		report("not", dhs.decrypt(a.zRep(), 0) == want);
	}
	{	// checkAnd, then a second multiplicative level on the result
		CuCtxt a, b, z; a.setLevel(0, 0, y0); b.setLevel(0, 0, y1);
		a.x2n(); b.x2n();
		cAnd(z, a, b); z.relin(); z.modSwitch();
		CuCtxt keep; copy(keep, z);
		z.x2z();
		const ZZX x01 = mulMod2(x0, x1, dhs.phi);
		report("and", dhs.decrypt(z.zRep(), 1) == x01);
		if (param.depth > 2) {
			CuCtxt c; c.setLevel(0, 0, y2); c.modSwitch();          // bring a fresh ciphertext to level 1
			c.x2n(); keep.x2n();
			CuCtxt w; cAnd(w, keep, c); w.relin(); w.modSwitch();
			w.x2z();
			report("and2", w.level() == 2 && dhs.decrypt(w.zRep(), 2) == mulMod2(x01, x2, dhs.phi));
		}
	}
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
