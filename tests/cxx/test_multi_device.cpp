// test_multi_device.cpp -- the reference's in-process multi-GPU mode (whole ciphertexts on device `dev`, one context
// per device, moveTo / copyTo between devices: cuhe/CuHE.cu:217-256, examples/Prince/Prince.cu:194-200) through
// the C++ API.  On a single-GPU box the devices are VIRTUAL (cuhe_hip_set_virtual_devices: every logical device has
// its own tables, keys, allocator and scratch but lives on the one physical GPU), which is what this test needs:
// the per-device bookkeeping, not the interconnect.  With >= 3 physical GPUs pass "real" to use them.
#include "CuHE.h"
#include "cuhe_hip.h"
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
using namespace cuHE;

static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } else printf("ok: %s\n", what); } while (0)
static ZZX randomPoly(int n, const ZZ &q) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, RandomBnd(q)); return r; }
static ZZX reduceCoeffs(const ZZX &a, const ZZ &q, int n) { ZZX r; for (int i = n - 1; i >= 0; --i) SetCoeff(r, i, coeff(a, i) % q); return r; }
static ZZX hostMul(const ZZX &a, const ZZX &b, const ZZX &phi, const ZZ &q, int n) { ZZX t = a * b; t %= phi; return reduceCoeffs(t, q, n); }

int main(int argc, char **argv) {
	const int NDEV = 3;
	if (!(argc > 1 && std::string(argv[1]) == "real")) cuhe_hip_set_virtual_devices(1);
	SetSeed(to_ZZ(31337));
	setParameters(3, 2, 8, 40, 20, 1155);
	multiGPUs(NDEV);
	CHECK(numGPUs() == NDEV, "multiGPUs");
	// Phi_1155 by repeated exact division (small ring: the host side of this test is schoolbook arithmetic)
	auto mu = [](int n) { int r = 1; for (int p = 2; p * p <= n; ++p) if (n % p == 0) { n /= p; if (n % p == 0) return 0; r = -r; } if (n > 1) r = -r; return r; };
	ZZX phi; SetCoeff(phi, 0, 1);
	const int m = 1155;
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == 1) { ZZX t; SetCoeff(t, 0, -1); SetCoeff(t, d, 1); phi *= t; }
	for (int d = 1; d <= m; ++d) if (m % d == 0 && mu(m / d) == -1) { ZZX t; SetCoeff(t, 0, -1); SetCoeff(t, d, 1); phi /= t; }
	std::vector<ZZ> q(param.depth);
	initCuHE(q.data(), phi);
	const int n = param.modLen;
	std::vector<ZZX> ek(param.numEvalKey);
	for (auto &e : ek) e = randomPoly(n, q[0]);
	initRelinearization(ek.data());                          // keys go to every device

	// the same product on every device
	ZZX a = randomPoly(n, q[0]), b = randomPoly(n, q[0]);
	const ZZX want = hostMul(a, b, phi, q[0], n);
	bool same = true;
	for (int dev = 0; dev < NDEV; ++dev) { ZZX c; mulZZX(c, a, b, 0, dev); same = same && c == want; }
	CHECK(same, "mulZZX gives the host product on every device");

	// a ciphertext made on device 0 travels 0 -> 1 -> 2, is multiplied and relinearised there, and comes back
	{
		CuCtxt x, y;
		x.setLevel(0, 0, a); y.setLevel(0, 2, b);
		x.x2n(); y.x2n();
		moveTo(x, 1);
		CHECK(x.device() == 1, "moveTo changes the device");
		CuCtxt xc; copyTo(xc, x, 2);
		CHECK(xc.device() == 2 && x.device() == 1, "copyTo leaves the source where it is");
		CuCtxt z; cAnd(z, xc, y);
		z.relin();
		CuCtxt back; copyTo(back, z, 0);
		z.x2z(); back.x2z();
		// the windowed key-switch sum on the host
		ZZX acc; const ZZ base = power2_ZZ(param.logRelin);
		for (int j = 0; j < param._numEvalKey(0); ++j) {
			ZZX win; const ZZ sh = power(base, j);
			for (int i = n - 1; i >= 0; --i) SetCoeff(win, i, (coeff(want, i) / sh) % base);
			acc += win * ek[j];
		}
		acc %= phi;
		const ZZX wantRelin = reduceCoeffs(acc, q[0], n);
		CHECK(z.zRep() == wantRelin && back.zRep() == wantRelin, "cAnd + relin on device 2 (keys resident there), result copied back to device 0");
	}
	// one host thread per device at the same time (the reference's OpenMP pattern, Prince.cu:194-200)
	{
		std::vector<ZZX> pa(NDEV), pb(NDEV), pc(NDEV);
		for (int d = 0; d < NDEV; ++d) { pa[d] = randomPoly(n, q[1]); pb[d] = randomPoly(n, q[1]); }
		std::vector<std::thread> th;
		for (int d = 0; d < NDEV; ++d) th.emplace_back([&, d] { for (int r = 0; r < 5; ++r) mulZZX(pc[d], pa[d], pb[d], 1, d); });
		for (auto &t : th) t.join();
		bool ok = true;
		for (int d = 0; d < NDEV; ++d) ok = ok && pc[d] == hostMul(pa[d], pb[d], phi, q[1], n);
		CHECK(ok, "one host thread per device, concurrently");
	}
	// one ciphertext with its CRT primes sharded over the devices (cAndRelinSharded): equal to cAnd + relin on one device,
	// at both levels that have enough primes, from device 0 and from device 1, several times in a row (buffer reuse)
	for (int lvl = 0; lvl <= 1; ++lvl)
		for (int dev0 = 0; dev0 <= 1; ++dev0) {
			ZZX pa = randomPoly(n, q[lvl]), pb = randomPoly(n, q[lvl]);
			CuCtxt x, y, z, w;
			x.setLevel(lvl, dev0, pa); y.setLevel(lvl, dev0, pb);
			x.x2n(); y.x2n();
			cAnd(z, x, y); z.relin(); z.x2z();
			bool ok = true;
			for (int r = 0; r < 3; ++r) { cAndRelinSharded(w, x, y); CuCtxt c; copy(c, w); c.x2z(); ok = ok && c.zRep() == z.zRep() && w.device() == dev0 && w.domain() == 2; }
			char what[128]; snprintf(what, sizeof what, "cAndRelinSharded over %d devices == cAnd + relin (level %d, home device %d)", NDEV, lvl, dev0);
			CHECK(ok, what);
		}
	printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
	return failures ? 1 : 0;
}
