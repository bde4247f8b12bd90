/*
 * oracle.h -- CPU restatement of the cuHE hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the MI355X backend.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may load it;
 * the product path (cuhe_amd/, include/) never links or calls anything here.
 *
 * Every function cites the reference file:line (relative to the upstream
 * vernamlab/cuHE tree) whose behaviour it restates.  Parity pin: the oracle is
 * checked against (1) the by-definition DFT of tests/test_ntt.cu:38-64,
 * (2) the L^-1 constants of cuhe/Base.cu:489,656,841, (3) mod-P arithmetic vs
 * big-integer arithmetic as tests/test_ModP.cu does, and (4) fixtures produced
 * by an independent pure-Python big-int script (tests/golden/gen_golden.py).
 * The reference itself is CUDA + NTL and cannot be built in this image, so
 * CRT / ICRT / Barrett / modswitch / relin are pinned against the Python
 * big-int fixtures and the ZZX-level identity (a*b mod Phi_m) mod q
 * (examples/DHS/DHS.cu:219-221), not against a reference binary.
 */
#ifndef CUHE_ORACLE_H
#define CUHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_P 0xffffffff00000001ULL          /* cuhe/ModP.h:33 */
#define ORC_G 15893793146607301539ULL        /* cuhe/Base.cu:65: primitive 65536-th root */
#define ORC_MAX_PRIMES 103                   /* cuhe/Base.cu:139 maxNumPrimes */
#define ORC_MAX_WORDS 112                    /* >= 103 primes * <=32 bits / 32, + slack */

/* OpenMP threads used by the per-prime / per-coefficient loops of the ctx stages below (default 1; 0 = all cores);
 * returns the count in effect.  Results are independent of it. */
int orc_set_threads(int n);

/* ---- field arithmetic mod P (cuhe/ModP.h:231-289, canonical results) ---- */
uint64_t orc_add_modP(uint64_t x, uint64_t y);
uint64_t orc_sub_modP(uint64_t x, uint64_t y);
uint64_t orc_mul_modP(uint64_t x, uint64_t y);
uint64_t orc_add_modP_div(uint64_t x, uint64_t y);   /* (x + y) % P and (x * y) % P by 128-bit division: cross-checks of the fold only */
uint64_t orc_mul_modP_div(uint64_t x, uint64_t y);
uint64_t orc_ls_modP(uint64_t x, int l);      /* x * 2^l mod P, any l >= 0 */
uint64_t orc_pow_modP(uint64_t x, uint64_t e);

/* ---- transforms (tests/test_ntt.cu:38-64; cuhe/Base.cu:309-842) ---- */
/* X[i] = sum_{j<len/2} x[j] * w^(i*j mod len) mod P, w = g^(65536/len). O(len^2). */
void orc_ntt_naive(uint64_t *dst, const uint32_t *src, int len);
/* same values, O(len log len) */
void orc_ntt_ext(uint64_t *dst, const uint32_t *src, int len);
/* `batch` independent orc_ntt_ext on `threads` OpenMP threads (0 = all); returns threads used */
int orc_ntt_ext_batch(uint64_t *dst, const uint32_t *src, int len, int batch, int threads);
/* the same transforms with per-length tables shared by the batch (the throughput form a host library would use; bench.py cpu_baseline) */
int orc_ntt_ext_fast_batch(uint64_t *dst, const uint32_t *src, int len, int batch, int threads);
/* full-length forward transform of u64 input (helper) */
void orc_ntt_full(uint64_t *dst, const uint64_t *src, int len);
/* x[j] = (len^-1 * sum_i X[i] w^(-ij) mod P) mod p  -- cuhe/Base.cu:438-490 */
void orc_intt_modp(uint32_t *dst, const uint64_t *src, int len, uint32_t p);
/* len^-1 mod P (cuhe/Base.cu:489,656,841) */
uint64_t orc_len_inv(int len);

/* ---- parameters (cuhe/Parameters.h:34-62, Parameters.cu:53-145) ---- */
typedef struct {
    int mSize, modLen, modLen2, rawLen, crtLen, nttLen;
    int logCoeffMax, logCoeffMin, logCoeffCut;
    int depth, modMsg, logMsg, wordsMsg;
    int logRelin, numEvalKey;
    int logCrtPrime, numCrtPrime;
} orc_params;

int orc_set_param(orc_params *q, int d, int p, int w, int min, int cut, int m);
int orc_num_crt_prime(const orc_params *q, int lvl);
int orc_log_coeff(const orc_params *q, int lvl);
int orc_words_coeff(const orc_params *q, int lvl);
int orc_num_eval_key(const orc_params *q, int lvl);
int orc_get_level(const orc_params *q, int logq);

/* deterministic primality for n < 2^32 */
int orc_is_prime_u32(uint32_t n);
/* cuhe/Operations.cu:37-80 */
int orc_gen_crt_primes(const orc_params *q, uint32_t *primes);
/* cyclotomic polynomial Phi_m, coefficients as int32, returns degree */
int orc_cyclotomic(int m, int32_t *coeffs, int cap);

/* ---- context: everything initCuHE precomputes (cuhe/CuHE.cu:36-50) ---- */
typedef struct orc_ctx orc_ctx;
/* modulus: monic integer polynomial of degree modLen, coefficient array of
 * length modLen+1 (NULL -> Phi_m). */
orc_ctx *orc_ctx_create(int d, int p, int w, int min, int cut, int m,
                        const int32_t *modulus);
void orc_ctx_destroy(orc_ctx *c);
const orc_params *orc_ctx_params(const orc_ctx *c);
const uint32_t *orc_ctx_primes(const orc_ctx *c);
/* coefficient modulus q_lvl = prod_{j<numCrtPrime-lvl} p_j as W LE words
 * (cuhe/Operations.cu:81-90); returns word count written */
int orc_ctx_coeff_modulus(const orc_ctx *c, int lvl, uint32_t *words, int cap);
/* p_i^-1 mod p_j, i>j, at [i*(i-1)/2+j] (cuhe/Operations.cu:91-99) */
const uint32_t *orc_ctx_invp(const orc_ctx *c);

/* ---- hot-path stages; lvl >= 0 ciphertext level ---- */
/* cuhe/Base.cu:857-879 : raw u32[rawLen][W] -> crt u32[np][crtLen] */
void orc_crt(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int lvl);
/* cuhe/Base.cu:880-924 : crt -> raw, value in [0, q_lvl) */
void orc_icrt(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int lvl);
/* cuhe/Operations.cu:394-398 : crt u32[np][crtLen] -> ntt u64[np][nttLen] */
void orc_ntt(const orc_ctx *c, uint64_t *dst, const uint32_t *src, int np);
/* cuhe/Operations.cu:405-411 inttHold: ntt -> u32[np][nttLen] */
void orc_intt_hold(const orc_ctx *c, uint32_t *dst, const uint64_t *src, int np);
/* cuhe/Operations.cu:419-427 intt: ntt -> crt u32[np][crtLen] (first crtLen) */
void orc_intt(const orc_ctx *c, uint32_t *dst, const uint64_t *src, int np);
/* exact f mod modulus per prime: in u32[np][nttLen] -> out u32[np][crtLen].
 * This is the meaning of cuhe/Operations.cu:460-501 (NTL: t %= polyMod). */
void orc_poly_reduce_exact(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int np);
/* step-by-step restatement of cuhe/Operations.cu:460-501 (Barrett via NTTs) */
void orc_barrett(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int np);
/* cuhe/Operations.cu:429-434 inttMod = inttHold + barrett */
void orc_intt_mod(const orc_ctx *c, uint32_t *dst, const uint64_t *src, int np);
/* cuhe/Base.cu:1036-1075 */
void orc_ntt_mul(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *y, int np);
void orc_ntt_add(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *y, int np);
void orc_ntt_mul_nx1(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *s, int np);
void orc_ntt_add_nx1(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *s, int np);
/* cuhe/Base.cu:1088-1109 */
void orc_crt_add(const orc_ctx *c, uint32_t *z, const uint32_t *x, const uint32_t *y, int np);
void orc_crt_add_int(const orc_ctx *c, uint32_t *z, const uint32_t *x, unsigned a, int np);
void orc_crt_add_nx1(const orc_ctx *c, uint32_t *z, const uint32_t *x, const uint32_t *s, int np);
/* cuhe/Base.cu:1112-1138 : u32[np][crtLen] -> u32[np-1][crtLen] */
void orc_modswitch(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int np);
/* cuhe/Base.cu:345-372 + Operations.cu:399-403 : raw -> u64[numEvalKey(lvl)][nttLen] */
void orc_nttw(const orc_ctx *c, uint64_t *dst, const uint32_t *raw, int lvl);
/* cuhe/Relinearization.cu:43-73 : evalkey raw u32[numEvalKey][rawLen][W0]
 * -> ek u64[numCrtPrime][numEvalKey][nttLen] (caller buffer) */
void orc_init_relin(const orc_ctx *c, uint64_t *ek, const uint32_t *evalkey_raw);
/* cuhe/Relinearization.cu:76-88 : dst u64[np][nttLen] */
void orc_relin(const orc_ctx *c, uint64_t *dst, const uint32_t *raw, int lvl, const uint64_t *ek);
/* cuhe/CuHE.cu:259-268 mulZZX at the raw level: out = (a*b mod Phi) mod q_lvl,
 * a,b,out raw u32[rawLen][W(lvl)] */
void orc_mul_raw(const orc_ctx *c, uint32_t *out, const uint32_t *a, const uint32_t *b, int lvl);
/* cAnd + relin (cuhe/CuHE.cu:101,570-581): operands in CRT domain at lvl,
 * result CRT domain u32[np][crtLen] */
void orc_mul_relin_crt(const orc_ctx *c, uint32_t *dst, const uint32_t *a, const uint32_t *b,
                       int lvl, const uint64_t *ek);

/* ---- products modulo x^n + 1 (rings with m = 2n a power of two): the remainder the reference's Barrett chain
 * (cuhe/Operations.cu:460-501) produces for Phi_m = x^n + 1, restated as the negacyclic convolution -- by definition
 * (O(n^2)) and through the twisted length-n transform (X[k] = sum_j x[j] psi^(j(2k+1)), psi a primitive 2n-th root of
 * unity, centred lift on the way back).  The fast forms return -1 when 2 n (p-1)^2 >= P. */
void orc_negacyclic_mul_modp_naive(uint32_t *dst, const uint32_t *a, const uint32_t *b, int n, uint32_t p);
int orc_negacyclic_mul_modp(uint32_t *dst, const uint32_t *a, const uint32_t *b, int n, uint32_t p);
void orc_nc_ntt(uint64_t *dst, const uint32_t *src, int n);
int orc_nc_intt_modp(uint32_t *dst, const uint64_t *src, int n, uint32_t p);
/* sum_j win[j] * key[j] mod (x^n + 1) mod p over k window / key rows of n coefficients (cuhe/Relinearization.cu:76-88) */
int orc_nc_relin_modp(uint32_t *dst, const uint32_t *win, const uint32_t *key, int k, int n, uint32_t p);
/* cAnd + relin of B pairs on x^n + 1 at config-4 sizes: per prime through the negacyclic restatement (CuHE.cu:101,570-581) */
int orc_nc_mul_relin_crt_batch(const orc_ctx *c, uint32_t *dst, const uint32_t *a, const uint32_t *b, int B, int lvl, const uint32_t *ekc);
/* the same chain for ONE pair with the keys transformed beforehand and table-sharing transforms: bench.py's CPU leg for mul + relin */
typedef struct orc_nc_prepared orc_nc_prepared;
orc_nc_prepared *orc_nc_prepare(const orc_ctx *c, int lvl, const uint32_t *ekc);        /* NULL: not a ring x^n + 1 within the lift bounds */
void orc_nc_prepared_free(orc_nc_prepared *P);
int orc_nc_mul_relin_prepared(const orc_nc_prepared *P, uint32_t *dst, const uint32_t *a, const uint32_t *b);

/* ---- optional second CPU baseline (bench.py): the product the reference delegates to NTL (examples/DHS/DHS.cu:219-221) on
 * x^n + 1 by Kronecker substitution and ONE big-integer multiplication through GMP, opened at run time (NTL is not in
 * this image; it builds on GMP).  orc_gmp_mul_xn1 returns -1 when no libgmp is found. */
int orc_gmp_available(void);
int orc_gmp_mul_xn1(uint32_t *out, const uint32_t *a, const uint32_t *b, int n, int W, const uint32_t *qwords, int qW);

/* seeded generator shared by tests / bench: splitmix64 */
uint64_t orc_splitmix64(uint64_t *state);
void orc_fill_u32_below(uint32_t *dst, size_t n, uint32_t bound, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
