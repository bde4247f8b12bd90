// test_mini_ntl.cpp -- CPU check of the header-only NTL subset (cuhe_amd/cxx/mini_ntl) that the C++ layer falls back to
// when NTL is not installed.  Prints one line per operation, "op operands... = result" in decimal; tests/test_mini_ntl.py
// recomputes every line with Python integers.  No GPU, no library: headers only.
#include <NTL/ZZ.h>
#include <NTL/ZZX.h>
#include <NTL/ZZ_p.h>
#include <NTL/ZZ_pX.h>
#include <NTL/ZZ_pE.h>
#include <iostream>
#include <vector>
NTL_CLIENT

static ZZ rnd(long bits, bool sign) { ZZ v = RandomBits_ZZ(bits); if (sign && IsOdd(RandomBits_ZZ(8))) v = -v; return v; }
static void poly(const char *tag, const ZZX &a) { std::cout << tag; for (long i = 0; i <= deg(a); ++i) std::cout << ' ' << coeff(a, i); std::cout << " ;"; }

int main() {
	SetSeed(to_ZZ(424242));
	const long sizes[] = {1, 31, 32, 33, 63, 64, 65, 96, 127, 128, 200, 521, 1000};
	for (long sa : sizes) for (long sb : sizes) {
		const ZZ a = rnd(sa, true), b = rnd(sb, true);
		std::cout << "add " << a << ' ' << b << " = " << a + b << '\n';
		std::cout << "sub " << a << ' ' << b << " = " << a - b << '\n';
		std::cout << "mul " << a << ' ' << b << " = " << a * b << '\n';
		if (!IsZero(b)) {
			const ZZ pa = a < to_ZZ(0) ? -a : a, pb = b < to_ZZ(0) ? -b : b;                 // the layer divides non-negative values only
			std::cout << "div " << pa << ' ' << pb << " = " << pa / pb << '\n';
			std::cout << "mod " << pa << ' ' << pb << " = " << pa % pb << '\n';
			std::cout << "gcd " << pa << ' ' << pb << " = " << GCD(pa, pb) << '\n';
		}
		std::cout << "cmp " << a << ' ' << b << " = " << (a < b ? -1 : (a == b ? 0 : 1)) << '\n';
	}
	for (long s : sizes) {
		const ZZ a = rnd(s, false);
		for (long k : {0L, 1L, 31L, 32L, 33L, 64L, 100L}) {
			std::cout << "shl " << a << ' ' << k << " = " << (a << k) << '\n';
			std::cout << "shr " << a << ' ' << k << " = " << (a >> k) << '\n';
		}
		std::cout << "bits " << a << " = " << NumBits(a) << '\n';
		unsigned char buf[160];
		BytesFromZZ(buf, a, 160);
		std::cout << "bytes " << a << " = " << ZZFromBytes(buf, 160) << '\n';
		BytesFromZZ(buf, a, 5);                                  // truncation to the low 5 bytes
		std::cout << "low40 " << a << " = " << ZZFromBytes(buf, 5) << '\n';
	}
	std::cout << "pow2 100 = " << power2_ZZ(100) << '\n';
	std::cout << "power 12345678901234567 7 = " << power(to_ZZ(12345678901234567L), 7) << '\n';
	// modular inverses
	for (int i = 0; i < 20; ++i) {
		ZZ n = rnd(90, false) + 3; ZZ a = RandomBnd(n);
		ZZ x; const long st = InvModStatus(x, a, n);
		std::cout << "invmod " << a << ' ' << n << " = " << (st ? to_ZZ(-1) : x) << '\n';
	}
	// ZZX: product, remainder by a monic polynomial
	for (int i = 0; i < 6; ++i) {
		ZZX a, b, m;
		const long da = 3 + 5 * i, db = 2 + 3 * i, dm = 4 + 2 * i;
		for (long j = 0; j <= da; ++j) SetCoeff(a, j, rnd(70, true));
		for (long j = 0; j <= db; ++j) SetCoeff(b, j, rnd(40, true));
		for (long j = 0; j < dm; ++j) SetCoeff(m, j, rnd(3, true));
		SetCoeff(m, dm, 1);
		poly("zzxmul", a); poly("", b); std::cout << " ="; poly("", a * b); std::cout << '\n';
		poly("zzxmod", a * b); poly("", m); std::cout << " ="; poly("", (a * b) % m); std::cout << '\n';
	}
	// ZZ_pE: inverse in Z_q[x]/(P) for a prime q and for a product of two primes (per-prime inverses + CRT lift)
	for (const char *qs : {"1048573", "4397987791019", "1099509530641"}) {   // a prime; 2097143 * 2097133 (two CRT primes of the DHS example); 197 * 5581266653 (a factor above 2^32)
		ZZ q = to_ZZ(0); for (const char *c = qs; *c; ++c) q = q * 10 + (*c - '0');
		ZZ_p::init(q);
		ZZX Pz; SetCoeff(Pz, 0, 1); SetCoeff(Pz, 1, 1); SetCoeff(Pz, 3, 1); SetCoeff(Pz, 4, 1); SetCoeff(Pz, 8, 1);      // x^8+x^4+x^3+x+1
		ZZ_pE::init(to_ZZ_pX(Pz));
		for (int i = 0; i < 4; ++i) {
			ZZX f; for (long j = 0; j < 8; ++j) SetCoeff(f, j, RandomBnd(q));
			const ZZ_pE e = to_ZZ_pE(to_ZZ_pX(f));
			const ZZ_pE g = inv(e);
			const ZZ_pE one = e * g;
			std::cout << "pEinv " << q << " ;"; poly("", f); std::cout << " ="; poly("", to_ZZX(rep(g))); poly("", to_ZZX(rep(one))); std::cout << '\n';
		}
	}
	return 0;
}
