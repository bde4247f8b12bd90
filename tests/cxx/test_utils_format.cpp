// test_utils_format.cpp -- the key-file text format of cuhe/Utils.h (Picklable / PicklableMap): known strings
// written by hand from the format rules of cuhe/Utils.cu:76-121,141-146,203-213, parse/print round trips, and
// the lookup behaviour examples/DHS/DHS.cu:62-124 depends on.  Host only.
#include "Utils.h"
#ifdef CUHE_MINI_NTL
#include <NTL/ZZ_pE.h>
#endif
#include <cstdio>
using namespace cuHE_Utils;

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } else printf("ok: %s\n", what); } while (0)

int main() {
	ZZX p;
	SetCoeff(p, 0, 3); SetCoeff(p, 2, 5); SetCoeff(p, 3, -7);
	Picklable pk("pk0", p);
	CHECK(pk.pickle() == "pk0,3,0,5,-7", "polynomial entry: key, then coefficients low degree first");
	CHECK(pk.getValues() == "3,0,5,-7" && pk.getKey() == "pk0" && pk.getCoeffsLen() == 4, "accessors");

	ZZ arr[3] = {to_ZZ(4), to_ZZ(0), to_ZZ(0)};
	Picklable d("d", arr, 3);
	CHECK(d.pickle() == "d,4", "an array goes through a polynomial: trailing zeros vanish");
	arr[0] = to_ZZ(9);
	CHECK(d.pickle() == "d,4" && d.getCoeffs()[0] == to_ZZ(4), "the entry keeps its own copy of the array");

	ZZ big = power2_ZZ(100) + to_ZZ(1);
	ZZ two[2] = {big, -big};
	Picklable cm("coeffMod", two, 2);
	CHECK(cm.pickle() == "coeffMod,1267650600228229401496703205377,-1267650600228229401496703205377", "multi-word integers in decimal");

	Picklable parsed("ek3,10,0,0,1267650600228229401496703205377,0");
	CHECK(parsed.getKey() == "ek3" && deg(parsed.getPoly()) == 3 && coeff(parsed.getPoly(), 3) == big && coeff(parsed.getPoly(), 0) == to_ZZ(10), "parse one entry");
	CHECK(parsed.pickle() == "ek3,10,0,0,1267650600228229401496703205377", "re-printing drops the trailing zero");
	Picklable gaps("k,,7,,8");
	CHECK(gaps.pickle() == "k,7,8", "empty fields vanish (strtok semantics)");

	Picklable semi("s;1;2;3", ";");
	CHECK(semi.getKey() == "s" && semi.pickle() == "s;1;2;3", "custom field separator");
	semi.setSeparator(" ");
	CHECK(semi.pickle() == "s 1 2 3", "setSeparator re-renders the values");

	Picklable copyOf(pk);
	CHECK(copyOf.pickle() == pk.pickle() && copyOf.getCoeffs() != pk.getCoeffs(), "copy owns its own coefficient array");

	vector<Picklable *> ps;
	ps.push_back(new Picklable("d", arr, 1));
	ps.push_back(new Picklable("polyMod", p));
	ps.push_back(new Picklable("pk0", p));
	PicklableMap m(ps);
	const string text = m.toString();
	CHECK(text == "d,9\npolyMod,3,0,5,-7\npk0,3,0,5,-7", "map: entries joined by newline, no trailing separator");
	PicklableMap back(text);
	CHECK(back.getPicklables().size() == 3 && back.toString() == text, "map text round trip");
	CHECK(back.get("polyMod")->getPoly() == p && back.get("d")->getValues() == "9", "lookup by key");
	bool threw = false;
	try { back.get("sk0"); } catch (char const *s) { threw = string(s) == "not found"; }
	CHECK(threw, "a missing key throws the C string \"not found\" (DHS.cu:85-90 catches char const*)");
	PicklableMap custom("a:1:2|b:3", "|", ":");
	CHECK(custom.get("b")->getValues() == "3" && custom.toString() == "a:1:2|b:3", "custom entry and field separators");

	// random round trip with signed multi-word coefficients
	SetSeed(to_ZZ(7));
	ZZX r;
	for (int i = 63; i >= 0; --i) SetCoeff(r, i, RandomBnd(power2_ZZ(200)) - power2_ZZ(199));
	Picklable rp("r", r);
	Picklable rq(rp.pickle());
	CHECK(rq.getPoly() == r, "random signed 200-bit coefficients survive print + parse");

#ifdef CUHE_MINI_NTL
	// the fallback big integer itself (cuhe_amd/cxx/mini_ntl): division identity on random and adversarial operands,
	// shifts, gcd, primality, polynomial helpers the DHS construction uses
	{
		int bad = 0;
		for (int it = 0; it < 5000; ++it) {
			const int ba = 1 + (int)(mini_next() % 700), bb = 1 + (int)(mini_next() % 700);
			ZZ a = RandomBits_ZZ(ba), b = RandomBits_ZZ(bb) + to_ZZ(1);
			if (it % 7 == 0) b = power2_ZZ(bb) - to_ZZ(1);
			if (it % 11 == 0) a = power2_ZZ(ba + 64) - to_ZZ(1);
			if (it % 3 == 0) a = -a;
			if (it % 5 == 0) b = -b;
			const ZZ q = a / b, rem = a % b;
			const bool ok = (q * b + rem == a) && ((b > to_ZZ(0)) ? (rem >= to_ZZ(0) && rem < b) : (rem <= to_ZZ(0) && rem > b));
			if (!ok) ++bad;
			if (!(((a << 37) >> 37) == a) || !((a << 5) == a * to_ZZ(32))) ++bad;
		}
		CHECK(bad == 0, "fallback ZZ: floor division identity, remainder sign, shifts");
		CHECK(GCD(84L, -36L) == 12 && GCD(to_ZZ(1) << 90, to_ZZ(3) << 70) == (to_ZZ(1) << 70), "fallback ZZ: GCD");
		CHECK(ProbPrime(2097143L) && ProbPrime(33554393L) && !ProbPrime(33554391L) && ProbPrime(18446744069414584321UL >> 1 ? 2147483647L : 2L) && !ProbPrime(1L), "fallback ZZ: ProbPrime");
		ZZX a, b; SetCoeff(a, 0, -1); SetCoeff(a, 15, 1);            // x^15 - 1
		SetCoeff(b, 0, -1); SetCoeff(b, 5, 1);                        // x^5 - 1
		ZZX qd = a / b, want; SetCoeff(want, 0, 1); SetCoeff(want, 5, 1); SetCoeff(want, 10, 1);
		CHECK(qd == want && (qd * b) == a && (a % b) == ZZX(), "fallback ZZX: exact division by a monic polynomial");
		CHECK((b * 3L) == (b * to_ZZ(3)) && coeff(b * 3L, 5) == to_ZZ(3) && coeff(b + 4L, 0) == to_ZZ(3), "fallback ZZX: scalar operations");
	}
	// modular polynomials of the fallback (what a DHS client needs: GF(2) batching arithmetic, the key inverse)
	{
		int bad = 0;
		ZZ_p::init(to_ZZ(2));
		for (int it = 0; it < 200; ++it) {
			ZZX az, bz; const int da = 1 + (int)(mini_next() % 300), db = 1 + (int)(mini_next() % 40);
			for (int i = 0; i <= da; ++i) SetCoeff(az, i, (long)(mini_next() & 1));
			for (int i = 0; i <= db; ++i) SetCoeff(bz, i, (long)(mini_next() & 1));
			SetCoeff(az, da, 1); SetCoeff(bz, db, 1);
			ZZ_pX a = to_ZZ_pX(az), b = to_ZZ_pX(bz), q, r;
			DivRem(q, r, a, b);
			if (!(q * b + r == a) || deg(r) >= deg(b) || !(to_ZZ_pX(to_ZZX(a)) == a)) ++bad;
		}
		ZZX pz; SetCoeff(pz, 0, 1); SetCoeff(pz, 1, 1); SetCoeff(pz, 4, 1);              // x^4 + x + 1, irreducible over GF(2)
		ZZ_pE::init(to_ZZ_pX(pz));
		ZZ_pX one; SetCoeff(one, 0, 1);
		for (int v = 1; v < 16; ++v) {
			ZZX fz; for (int i = 0; i < 4; ++i) SetCoeff(fz, i, (long)((v >> i) & 1));
			const ZZ_pE f = to_ZZ_pE(to_ZZ_pX(fz));
			if (!(rep(f * inv(f)) == one)) ++bad;
		}
		CHECK(bad == 0, "fallback ZZ_pX over GF(2): division identity, inverses in GF(16)");
		bad = 0;
		const ZZ q3 = to_ZZ(2097143L) * to_ZZ(2097133L) * to_ZZ(524287L);              // CRT primes of the (5,2,1,61,20,8191) set
		ZZ_p::init(q3);
		ZZX cz; SetCoeff(cz, 0, 1); SetCoeff(cz, 16, 1);
		ZZ_pE::init(to_ZZ_pX(cz));
		ZZ_pX one3; SetCoeff(one3, 0, 1);
		int inverted = 0;
		for (int it = 0; it < 10; ++it) {
			ZZX fz; for (int i = 0; i < 16; ++i) SetCoeff(fz, i, RandomBnd(q3));
			const ZZ_pE f = to_ZZ_pE(to_ZZ_pX(fz));
			try { if (!(rep(f * inv(f)) == one3)) ++bad; ++inverted; } catch (std::runtime_error &) {}
		}
		SetCoeff(cz, 0, 0);                                                             // x^16: x itself is not a unit
		ZZ_pE::init(to_ZZ_pX(cz));
		ZZX xz; SetCoeff(xz, 1, 1);
		bool threw = false;
		try { inv(to_ZZ_pE(to_ZZ_pX(xz))); } catch (std::runtime_error &) { threw = true; }
		CHECK(bad == 0 && inverted >= 8 && threw, "fallback ZZ_pE over a composite modulus: per-prime inverse + CRT lift; runtime_error when there is none");
	}
#endif
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
