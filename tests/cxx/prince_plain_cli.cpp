// prince_plain_cli.cpp -- prints what tests/cxx/prince_common.hpp computes, for the CPU test tests/test_prince_plain.py:
//   prince_plain_cli <pt hex> <k0 hex> <k1 hex>   ->  ciphertext, then the state after each of the 12 S-box layers
#include "prince_common.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) {
	if (argc != 4) return 2;
	std::vector<u64x> st;
	const u64x ct = plainPrince(strtoull(argv[1], 0, 16), strtoull(argv[2], 0, 16), strtoull(argv[3], 0, 16), &st);
	printf("%016llx\n", ct);
	for (u64x s : st) printf("%016llx\n", s);
	// the S-box circuits are derived from the table: check that the derived normal forms evaluate back to the table
	int inv[16]; for (int i = 0; i < 16; ++i) inv[SBOX[i]] = i;
	for (const int *box : {SBOX, (const int *)inv}) {
		const Anf f = anfOf(box);
		for (int x = 0; x < 16; ++x) {
			int y = 0;
			for (int o = 0; o < 4; ++o) { int b = 0; for (int m = 0; m < 16; ++m) if (f.c[o][m] && (x & m) == m) b ^= 1; y = (y << 1) | b; }
			if (y != box[x]) { printf("anf mismatch\n"); return 1; }
		}
		for (int o = 0; o < 4; ++o) if (f.c[o][15]) { printf("degree 4 term\n"); return 1; }
	}
	printf("anf ok\n");
	return 0;
}
