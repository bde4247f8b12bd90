// dhs_client.hpp -- a small DHS / LTV scheme client written from scratch on top of the C++ drop-in API
// (cuhe_amd/cxx/CuHE.h), shared by test_dhs_flow.cpp and test_prince_flow.cpp.  It stands where the
// reference's examples/DHS/DHS.cu (class CuDHS) stands; that file needs NTL's ZZ_pE / ZZ_pX machinery, which
// the image lacks, so the client here restates the SCHEME (DHS.cu:212-372) with its own number theory:
//     f = 2f'+1 invertible in Z_q0[x]/Phi_m,  pk = 2 g f^-1,  Enc(m) = pk s + 2e + m,
//     Dec(c) = centred(f c) mod 2,  ek_j = pk s_j + 2 e_j + f 2^(w j),  all samples ternary (B = 1).
// Every ring product goes through mulZZX, i.e. through the GPU hot path; the host does additions of small
// values only (no big-integer division anywhere, so the fallback big integer is fast enough for n = 16384).
#pragma once
#include "CuHE.h"
#include "cuhe_hip.h"
#include <cstdlib>
// CUHE_TRACE_MARK=1: a recognisable kernel (the library's VALU probe, ~1 ms) before and after the timed part of a client, so that a
// rocprofv3 kernel trace can be cut to that part (tools/rocpd_summary.py --between k_probe_valu)
static inline void traceMark() {
	if (!getenv("CUHE_TRACE_MARK")) return;
	double a, b, c;
	cuhe_hip_device_sync(0);
	cuhe_hip_probe_valu(0, 1, 1, &a, &b, &c);
}
#include "Utils.h"
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace dhs_client {
using namespace cuHE;
typedef long long i64;
typedef unsigned long long u64x;

inline std::vector<i64> cyclotomicInts(int m) {
	auto mu = [](int n) { int r = 1; for (int p = 2; p * p <= n; ++p) if (n % p == 0) { n /= p; if (n % p == 0) return 0; r = -r; } if (n > 1) r = -r; return r; };
	std::vector<i64> a(2 * (size_t)m + 2, 0); int len = 1; a[0] = 1;
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == 1) { for (int i = len - 1; i >= 0; --i) { a[i + d] += a[i]; a[i] = -a[i]; } len += d; }
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == -1) { for (int i = 0; i < len - d; ++i) a[i] = (i >= d ? a[i - d] : 0) - a[i]; len -= d; }
	a.resize(len); return a;
}

inline u64x powmod(u64x b, u64x e, u64x p) { u64x r = 1; b %= p; while (e) { if (e & 1) r = r * b % p; b = b * b % p; e >>= 1; } return r; }

// (a + c*b) mod p for a, b, c in [0, p), p < 2^26: exact in double precision (c*b + a < 2^53)
static inline double fmaMod(double a, double c, double b, double p, double pinv) {
	const double x = c * b + a;
	const double q = (x * pinv + 6755399441055744.0) - 6755399441055744.0;      // round to nearest integer
	double r = x - q * p;
	if (r < 0) r += p;
	if (r >= p) r -= p;
	return r;
}
// inverse of f modulo (phi, p) over F_p by the extended Euclidean algorithm; false if gcd(f, phi) != 1
inline bool invertModPrime(std::vector<unsigned> &inv, const std::vector<unsigned> &f, const std::vector<unsigned> &phi, unsigned p) {
	typedef std::vector<double> Poly;
	const double P = p, Pinv = 1.0 / p;
	auto trim = [](Poly &a) { while (!a.empty() && a.back() == 0) a.pop_back(); };
	Poly r0(phi.begin(), phi.end()), r1(f.begin(), f.end()), t0, t1(1, 1.0);
	trim(r0); trim(r1);
	while (!r1.empty()) {
		const double lead = (double)powmod((u64x)r1.back(), p - 2, p);
		while (r0.size() >= r1.size()) {            // r0 -= c x^sh r1,  t0 -= c x^sh t1
			const size_t sh = r0.size() - r1.size();
			const double c = fmaMod(0, r0.back(), lead, P, Pinv), nc = c == 0 ? 0 : P - c;
			double *a = r0.data() + sh; const double *b = r1.data();
			for (size_t i = 0, e = r1.size(); i < e; ++i) a[i] = fmaMod(a[i], nc, b[i], P, Pinv);
			if (t0.size() < t1.size() + sh) t0.resize(t1.size() + sh, 0);
			a = t0.data() + sh; b = t1.data();
			for (size_t i = 0, e = t1.size(); i < e; ++i) a[i] = fmaMod(a[i], nc, b[i], P, Pinv);
			trim(r0);
			if (r0.empty()) break;
		}
		std::swap(r0, r1); std::swap(t0, t1);
	}
	if (r0.size() != 1) return false;
	const double g = (double)powmod((u64x)r0[0], p - 2, p);
	Poly t = t0; const size_t n = phi.size() - 1;
	for (size_t k = t.size(); k-- > n;) {           // reduce modulo phi (monic)
		const double c = t[k]; if (c == 0) continue;
		const double nc = P - c;
		for (size_t i = 0; i <= n; ++i) t[k - n + i] = fmaMod(t[k - n + i], nc, (double)phi[i], P, Pinv);
	}
	inv.assign(n, 0);
	for (size_t i = 0; i < n && i < t.size(); ++i) inv[i] = (unsigned)fmaMod(0, t[i], g, P, Pinv);
	return true;
}

struct Rng {             // xorshift64*: the client's own sampler (ternary noise, message bits)
	u64x s;
	explicit Rng(u64x seed) : s(seed * 0x9E3779B97F4A7C15ULL + 1) {}
	u64x next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1DULL; }
	int ternary() { return (int)(next() % 3) - 1; }
	int bit() { return (int)(next() >> 63); }
};

struct Dhs {
	int n = 0, depth = 0, np = 0;
	std::vector<ZZ> q;                 // coefficient modulus per level (initCuHE's output)
	std::vector<unsigned> primes;
	std::vector<i64> phi;
	ZZX phiZ;
	std::vector<ZZX> pk, sk, ek;
	std::vector<int> fSmall;           // the secret key's small signed coefficients
	Rng rng{12345};

	std::vector<int> sampleSmall() { std::vector<int> v(n); for (auto &x : v) x = rng.ternary(); return v; }
	// small signed coefficients -> residues in [0, q)
	static ZZX liftSmall(const std::vector<int> &v, const ZZ &q) {
		ZZX r;
		for (long i = (long)v.size() - 1; i >= 0; --i) if (v[i]) SetCoeff(r, i, v[i] > 0 ? to_ZZ((long)v[i]) : q + to_ZZ((long)v[i]));
		return r;
	}
	// (a + small) mod q for a in [0, q), |small| tiny
	static ZZ addSmall(const ZZ &a, long small, const ZZ &q) {
		ZZ r = a + to_ZZ(small);
		if (r < to_ZZ(0)) r += q; else if (r >= q) r -= q;
		return r;
	}
	// residues mod q  ->  residues mod a divisor q' of q that is not much smaller per step (levels)
	static ZZX reduceTo(const ZZX &a, const ZZ &qto) { ZZX r; for (long i = deg(a); i >= 0; --i) SetCoeff(r, i, coeff(a, i) % qto); return r; }

	bool invert(ZZX &finv, const std::vector<int> &f) {
		// Z_q0[x]/Phi is the product of the F_p[x]/Phi over the CRT primes: invert per prime, then lift
		std::vector<std::vector<unsigned>> rows(np);
		std::vector<char> ok(np, 0);
		std::vector<std::thread> pool;
		for (int i = 0; i < np; ++i) pool.emplace_back([&, i] {
			const unsigned p = primes[i];
			std::vector<unsigned> fp(n), php(n + 1);
			for (int k = 0; k < n; ++k) fp[k] = (unsigned)((f[k] % (int)p + (int)p) % (int)p);
			for (int k = 0; k <= n; ++k) php[k] = (unsigned)(((phi[k] % (i64)p) + p) % p);
			ok[i] = invertModPrime(rows[i], fp, php, p);
		});
		for (auto &t : pool) t.join();
		for (int i = 0; i < np; ++i) if (!ok[i]) return false;
		std::vector<ZZ> lift(np);
		for (int i = 0; i < np; ++i) {
			const ZZ mi = q[0] / to_ZZ((long)primes[i]);
			const u64x bi = powmod((u64x)to_long(mi % to_ZZ((long)primes[i])), primes[i] - 2, primes[i]);
			lift[i] = (mi * to_ZZ((long)bi)) % q[0];
		}
		clear(finv);
		for (int k = n - 1; k >= 0; --k) {
			ZZ v;
			for (int i = 0; i < np; ++i) v += lift[i] * to_ZZ((long)rows[i][k]);
			SetCoeff(finv, k, v % q[0]);
		}
		return true;
	}

	// pk * s + 2 e + extra   (mod q_lvl), s and e fresh ternary samples
	ZZX maskedSample(int lvl, const ZZX &extra) {
		ZZX t;
		mulZZX(t, pk[lvl], liftSmall(sampleSmall(), q[lvl]), lvl, 0, 0);
		const std::vector<int> e = sampleSmall();
		ZZX r;
		for (int i = n - 1; i >= 0; --i) {
			ZZ v = addSmall(coeff(t, i), 2L * e[i], q[lvl]) + coeff(extra, i);
			if (v >= q[lvl]) v -= q[lvl];
			SetCoeff(r, i, v);
		}
		return r;
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
		ZZX finv;
		for (;;) {
			fSmall = sampleSmall();
			for (auto &x : fSmall) x *= param.modMsg;
			fSmall[0] += 1;
			if (invert(finv, fSmall)) break;
		}
		pk.resize(depth); sk.resize(depth);
		sk[0] = liftSmall(fSmall, q[0]);
		ZZX gf;
		mulZZX(gf, liftSmall(sampleSmall(), q[0]), finv, 0, 0, 0);
		for (int i = n - 1; i >= 0; --i) { ZZ v = coeff(gf, i) * to_ZZ((long)param.modMsg); while (v >= q[0]) v -= q[0]; SetCoeff(pk[0], i, v); }
		for (int i = 1; i < depth; ++i) { sk[i] = liftSmall(fSmall, q[i]); pk[i] = reduceTo(pk[i - 1], q[i]); }
		// evaluation keys (DHS.cu:323-345): ek_j = pk s + 2e + f 2^(w j)
		if (param.logRelin > 0) {
			ek.resize(param.numEvalKey);
			for (int j = 0; j < param.numEvalKey; ++j) {
				ZZ tw = power2_ZZ((long)param.logRelin * j);
				while (tw >= q[0]) tw -= q[0];
				ZZX tp;
				for (int k = n - 1; k >= 0; --k) {
					const int c = fSmall[k];
					if (!c) continue;
					ZZ v = tw * to_ZZ((long)(c < 0 ? -c : c));
					while (v >= q[0]) v -= q[0];
					if (c < 0 && !IsZero(v)) v = q[0] - v;
					SetCoeff(tp, k, v);
				}
				ek[j] = maskedSample(0, tp);
			}
			initRelinearization(ek.data());
		}
	}
	// ---- key strings (examples/DHS/DHS.cu:120-189: getPublicKey / getPrivateKey through cuhe/Utils.h's PicklableMap) and the scheme
	// object built back from one (DHS.cu:57-118: setParameters + initCuHE AGAIN in the same process; no key generation).  A public
	// string gives an object that encrypts and evaluates, a private one also decrypts.
	std::vector<cuHE_Utils::Picklable *> publicPicklables() {
		using cuHE_Utils::Picklable;
		std::vector<Picklable *> ps;
		const long ints[6] = {param.depth, param.modMsg, param.logRelin, param.logCoeffMin, param.logCoeffCut, param.mSize};
		const char *names[6] = {"d", "p", "w", "min", "cut", "m"};
		for (int i = 0; i < 6; ++i) { ZZ v = to_ZZ(ints[i]); ps.push_back(new Picklable(names[i], &v, 1)); }
		ps.push_back(new Picklable("coeffMod", q.data(), depth));
		ps.push_back(new Picklable("polyMod", phiZ));
		for (int i = 0; i < depth; ++i) ps.push_back(new Picklable("pk" + std::to_string(i), pk[i]));
		for (size_t i = 0; i < ek.size(); ++i) ps.push_back(new Picklable("ek" + std::to_string(i), ek[i]));
		return ps;
	}
	std::string getPublicKey() { cuHE_Utils::PicklableMap pm(publicPicklables()); return pm.toString(); }
	std::string getPrivateKey() {
		std::vector<cuHE_Utils::Picklable *> ps = publicPicklables();
		for (int i = 0; i < depth; ++i) ps.push_back(new cuHE_Utils::Picklable("sk" + std::to_string(i), sk[i]));
		cuHE_Utils::PicklableMap pm(ps);
		return pm.toString();
	}
	bool hasPrivate() const { return !sk.empty(); }
	// reloadKeys: run initRelinearization on the string's evaluation keys too (the reference's constructor does not: its second
	// and third objects live on the keys the first one loaded)
	void setupFromKey(const std::string &key, bool reloadKeys) {
		cuHE_Utils::PicklableMap pm(key);
		auto num = [&](const char *k) { return atoi(pm.get(k)->getValues().c_str()); };
		setParameters(num("d"), num("p"), num("w"), num("min"), num("cut"), num("m"));
		n = param.modLen; depth = param.depth; np = param.numCrtPrime;
		q.assign(depth, ZZ());
		{ cuHE_Utils::Picklable *cm = pm.get("coeffMod"); for (int i = 0; i < depth && i < cm->getCoeffsLen(); ++i) q[i] = cm->getCoeffs()[i]; }
		phiZ = pm.get("polyMod")->getPoly();
		phi.assign(n + 1, 0);
		for (int i = 0; i <= n; ++i) { long v; conv(v, coeff(phiZ, i)); phi[i] = v; }
		pk.assign(depth, ZZX());
		for (int i = 0; i < depth; ++i) pk[i] = pm.get("pk" + std::to_string(i))->getPoly();
		sk.clear();
		try { pm.get("sk0"); sk.assign(depth, ZZX()); } catch (char const *) {}
		for (size_t i = 0; i < sk.size(); ++i) sk[i] = pm.get("sk" + std::to_string((int)i))->getPoly();
		ek.clear();
		if (param.logRelin > 0) {
			ek.assign(param.numEvalKey, ZZX());
			for (int i = 0; i < param.numEvalKey; ++i) ek[i] = pm.get("ek" + std::to_string(i))->getPoly();
		}
		std::vector<ZZ> fromLibrary(depth);
		initCuHE(fromLibrary.data(), phiZ);
		for (int i = 0; i < depth; ++i) if (!(fromLibrary[i] == q[i])) { printf("key string: coefficient modulus of level %d differs from the library's\n", i); exit(2); }
		primes.resize(np);
		if (cuhe_hip_get_crt_primes(primes.data(), np) != 0) { printf("cannot read the CRT primes\n"); exit(2); }
		if (reloadKeys && !ek.empty()) initRelinearization(ek.data());
	}
	ZZX encrypt(const ZZX &msg, int lvl) { return maskedSample(lvl, msg); }
	ZZX encryptBit(int bit, int lvl) { ZZX m; if (bit) SetCoeff(m, 0, 1); return encrypt(m, lvl); }
	// centred(f c) mod 2, coefficient by coefficient (q_lvl is odd: subtracting it flips the parity)
	ZZX decrypt(const ZZX &c, int lvl) {
		ZZX t, out;
		mulZZX(t, c, sk[lvl], lvl, 0, 0);
		const ZZ half = (q[lvl] - to_ZZ(1)) / to_ZZ(2);
		for (long i = deg(t); i >= 0; --i) {
			const ZZ &x = coeff(t, i);
			if (IsZero(x)) continue;
			const int low = (int)IsOdd(x);
			const int par = (x > half) ? (low ^ 1) : low;
			if (par) SetCoeff(out, i, 1);
		}
		return out;
	}
};

} // namespace dhs_client
