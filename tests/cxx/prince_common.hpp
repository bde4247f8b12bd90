// prince_common.hpp -- the PRINCE block cipher on bit vectors and the algebraic normal form of its S-boxes, shared
// by the two homomorphic evaluations (test_prince_flow.cpp: CuHE.h gates; test_prince_batched.cpp: gates on arrays).
// Written from the published specification (Borghoff et al., ASIACRYPT 2012); both programs first check
// plainPrince against the five test vectors of that paper.
#pragma once
#include <vector>
typedef unsigned long long u64x;

// ------------------------------------------------------------------ plain PRINCE on bit vectors (bit 0 = MSB)
static const int SBOX[16] = {0xB, 0xF, 0x3, 0x2, 0xA, 0xC, 0x9, 0x1, 0x6, 0x7, 0x8, 0x0, 0xE, 0x5, 0xD, 0x4};
static const u64x RC[12] = {0x0000000000000000ULL, 0x13198a2e03707344ULL, 0xa4093822299f31d0ULL, 0x082efa98ec4e6c89ULL,
                            0x452821e638d01377ULL, 0xbe5466cf34e90c6cULL, 0x7ef84f78fd955cb1ULL, 0x85840851f1ac43aaULL,
                            0xc882d32f25323c54ULL, 0x64a51195e0e3610dULL, 0xd3b5a399ca0c2399ULL, 0xc0ac29b7c97c50ddULL};
static const int SR[16] = {0, 5, 10, 15, 4, 9, 14, 3, 8, 13, 2, 7, 12, 1, 6, 11};
typedef std::vector<int> Bits;
static Bits toBits(u64x x) { Bits b(64); for (int i = 0; i < 64; ++i) b[i] = (int)((x >> (63 - i)) & 1); return b; }
static u64x toVal(const Bits &b) { u64x v = 0; for (int i = 0; i < 64; ++i) v = (v << 1) | (u64x)b[i]; return v; }
// M' as a list of three source bits per output bit: block-diagonal (M^0, M^1, M^1, M^0), each 16x16 built from the
// 4x4 identity matrices with one diagonal entry cleared
static std::vector<std::vector<int>> mPrimeSources() {
	std::vector<std::vector<int>> src(64);
	const int first[4] = {0, 1, 1, 0};
	for (int chunk = 0; chunk < 4; ++chunk)
		for (int br = 0; br < 4; ++br) for (int r = 0; r < 4; ++r)
			for (int bc = 0; bc < 4; ++bc) {
				const int cleared = (first[chunk] + br + bc) % 4;
				if (r != cleared) src[16 * chunk + 4 * br + r].push_back(16 * chunk + 4 * bc + r);
			}
	return src;
}
static Bits plainSub(const Bits &b, const int *box) {
	Bits o(64);
	for (int i = 0; i < 16; ++i) {
		const int v = box[(b[4 * i] << 3) | (b[4 * i + 1] << 2) | (b[4 * i + 2] << 1) | b[4 * i + 3]];
		for (int k = 0; k < 4; ++k) o[4 * i + k] = (v >> (3 - k)) & 1;
	}
	return o;
}
static Bits plainMPrime(const Bits &b) { static const auto src = mPrimeSources(); Bits o(64); for (int i = 0; i < 64; ++i) { int v = 0; for (int s : src[i]) v ^= b[s]; o[i] = v; } return o; }
static Bits plainSR(const Bits &b) { Bits o(64); for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) o[4 * i + k] = b[4 * SR[i] + k]; return o; }
static Bits plainSRinv(const Bits &b) { Bits o(64); for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) o[4 * SR[i] + k] = b[4 * i + k]; return o; }
static Bits plainXor(Bits a, const Bits &b) { for (int i = 0; i < 64; ++i) a[i] ^= b[i]; return a; }
static u64x plainPrince(u64x pt, u64x k0, u64x k1, std::vector<u64x> *states) {
	int inv[16]; for (int i = 0; i < 16; ++i) inv[SBOX[i]] = i;
	const u64x k0p = ((k0 >> 1) | ((k0 & 1) << 63)) ^ (k0 >> 63);
	Bits s = plainXor(plainXor(plainXor(toBits(pt), toBits(k0)), toBits(k1)), toBits(RC[0]));
	for (int i = 1; i <= 5; ++i) {
		s = plainSub(s, SBOX); if (states) states->push_back(toVal(s));
		s = plainXor(plainXor(plainSR(plainMPrime(s)), toBits(RC[i])), toBits(k1));
	}
	s = plainSub(s, SBOX); if (states) states->push_back(toVal(s));
	s = plainSub(plainMPrime(s), inv); if (states) states->push_back(toVal(s));
	for (int i = 6; i <= 10; ++i) {
		s = plainXor(plainXor(s, toBits(k1)), toBits(RC[i]));
		s = plainSub(plainMPrime(plainSRinv(s)), inv); if (states) states->push_back(toVal(s));
	}
	return toVal(plainXor(plainXor(plainXor(s, toBits(RC[11])), toBits(k1)), toBits(k0p)));
}

// algebraic normal form of a 4-bit S-box: anf[o][mask] = coefficient of the monomial `mask` (8 = a, 4 = b, 2 = c, 1 = d;
// a is the most significant input bit) in output bit o (0 = most significant)
struct Anf { int c[4][16]; };
static Anf anfOf(const int *box) {
	Anf f;
	for (int o = 0; o < 4; ++o) {
		int t[16]; for (int x = 0; x < 16; ++x) t[x] = (box[x] >> (3 - o)) & 1;
		for (int bit = 1; bit < 16; bit <<= 1) for (int x = 0; x < 16; ++x) if (x & bit) t[x] ^= t[x ^ bit];
		for (int x = 0; x < 16; ++x) f.c[o][x] = t[x];
	}
	return f;
}

