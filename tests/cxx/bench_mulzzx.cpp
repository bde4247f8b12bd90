// bench_mulzzx.cpp -- wall-clock of the NTL-facing full multiply mulZZX (cuhe/CuHE.cu:259-268) through the
// C++ drop-in API, with the host staging (ZZX <-> raw words, PCIe) that SURVEY 8(f3) says dominates once the
// kernels are fast, split by stage.  BASELINE config 3 by default: N = 2^15, 32 CRT primes.
// usage: bench_mulzzx [reps] [d p w min cut m]
#include "CuHE.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace cuHE;
typedef std::chrono::steady_clock clk;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char **argv) {
	int reps = argc > 1 ? atoi(argv[1]) : 10;
	int prm[6] = {9, 2, 16, 576, 24, 65536};
	if (argc == 8) for (int i = 0; i < 6; ++i) prm[i] = atoi(argv[i + 2]);
	setParameters(prm[0], prm[1], prm[2], prm[3], prm[4], prm[5]);
	ZZX phi; SetCoeff(phi, 0, 1); SetCoeff(phi, param.modLen, 1);           // x^n + 1 (m a power of two)
	std::vector<ZZ> q(param.depth);
	initCuHE(q.data(), phi);
	const int n = param.modLen;
	SetSeed(to_ZZ(42));
	ZZX a, b, c;
	for (int i = n - 1; i >= 0; --i) { SetCoeff(a, i, RandomBnd(q[0])); SetCoeff(b, i, RandomBnd(q[0])); }
	printf("mulZZX: n=%d nttLen=%d primes=%d words/coeff=%d\n", n, param.nttLen, param.numCrtPrime, param._wordsCoeff(0));
	for (int alloc = 0; alloc < 2; ++alloc) {
		if (alloc) startAllocator();
		mulZZX(c, a, b, 0, 0);                                              // warm-up
		std::vector<double> t(reps);
		for (int r = 0; r < reps; ++r) { auto t0 = clk::now(); mulZZX(c, a, b, 0, 0); t[r] = ms(t0, clk::now()); }
		std::sort(t.begin(), t.end());
		printf("allocator %s: mulZZX median %.3f ms  min %.3f ms\n", alloc ? "on " : "off", t[reps / 2], t[0]);
		// stage split through the public state machine
		double s[8] = {0};
		for (int r = 0; r < reps; ++r) {
			CuCtxt x, y;
			auto t0 = clk::now(); x.setLevel(0, 0, a); y.setLevel(0, 0, b);
			auto t1 = clk::now(); x.x2r(); y.x2r();
			auto t2 = clk::now(); x.x2c(); y.x2c();
			auto t3 = clk::now(); x.x2n(); y.x2n();
			auto t4 = clk::now(); cAnd(x, x, y);
			auto t5 = clk::now(); x.x2c();
			auto t6 = clk::now(); x.x2r();
			auto t7 = clk::now(); x.x2z();
			auto t8 = clk::now();
			const clk::time_point tp[9] = {t0, t1, t2, t3, t4, t5, t6, t7, t8};
			for (int i = 0; i < 8; ++i) s[i] += ms(tp[i], tp[i + 1]);
		}
		const char *names[8] = {"setLevel(ZZX copy) x2", "z2r (pack+H2D) x2", "r2c (crt) x2", "c2n (ntt) x2", "cAnd", "n2c (intt+reduce)", "c2r (icrt)", "r2z (D2H+unpack)"};
		for (int i = 0; i < 8; ++i) printf("    %-24s %8.3f ms\n", names[i], s[i] / reps);
	}
	stopAllocator();
	return 0;
}
