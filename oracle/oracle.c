/*
 * oracle.c -- CPU restatement of the cuHE hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h).  Plain C99 + unsigned __int128; no external dependencies.
 *
 * Parity pin status: transforms / field arithmetic are pinned by the
 * reference's own by-definition test (tests/test_ntt.cu:38-64) and constants;
 * CRT/ICRT/Barrett/modswitch/relin are pinned against the independent
 * pure-Python big-int fixtures in tests/golden/ (the reference holds no
 * vectors for them and cannot be built here: CUDA + NTL).
 */
#include "oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* OpenMP threads for the loops over CRT primes / coefficients of the ctx stages (1 = serial, the default; bench.py's CPU
 * baseline sets all host cores: "OpenMP over primes").  Results do not depend on it. */
static int g_threads = 1;
int orc_set_threads(int n) {
#ifdef _OPENMP
    if (n <= 0) n = omp_get_max_threads();
    g_threads = n;
#else
    (void)n; g_threads = 1;
#endif
    return g_threads;
}

/* ------------------------------------------------------------------------ */
/* field arithmetic mod P = 2^64 - 2^32 + 1      (cuhe/ModP.h:231-289)       */
/* ------------------------------------------------------------------------ */
/* Reduction of a 128-bit value the way the reference's field arithmetic does it (ModP.h:249-289: the product is folded with
 * 2^64 = 2^32 - 1 and 2^96 = -1 mod P, no division): x = lo + 2^64 hl + 2^96 hh  =  lo + (2^32 - 1) hl - hh  (mod P).
 * Canonical result (< P) for EVERY 128-bit input.  Round 6: this replaces `u128 % P` (libgcc __umodti3, ~40 ns) in every
 * function below -- same values, checked against the division on random and edge inputs by tests/test_oracle_golden.py through
 * orc_mul_modP_div / orc_add_modP_div -- so that the oracle timed as bench.py's cpu_baseline is "the same algorithm on the host
 * cores" rather than a benchmark of 128-bit division. */
static inline uint64_t fold128(u128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFu;
    uint64_t t = lo - hh;
    if (lo < hh) t -= 0xFFFFFFFFu;                       /* wrapped by 2^64 = 2^32 - 1 (mod P): take it off again */
    const uint64_t m = hl * 0xFFFFFFFFu;                 /* < 2^64 */
    uint64_t r = t + m;
    if (r < m) r += 0xFFFFFFFFu;                         /* carry out: + 2^64 = + 2^32 - 1 */
    return r >= ORC_P ? r - ORC_P : r;
}
uint64_t orc_add_modP_div(uint64_t x, uint64_t y) { return (uint64_t)(((u128)x + y) % ORC_P); }     /* the division forms: cross-checks only */
uint64_t orc_mul_modP_div(uint64_t x, uint64_t y) { return (uint64_t)(((u128)x * y) % ORC_P); }
uint64_t orc_add_modP(uint64_t x, uint64_t y) {          /* ModP.h:231-239 */
    return fold128((u128)x + y);
}
uint64_t orc_sub_modP(uint64_t x, uint64_t y) {          /* ModP.h:241-247 */
    if (x >= ORC_P) x -= ORC_P;
    if (y >= ORC_P) y -= ORC_P;
    return x >= y ? x - y : x + (ORC_P - y);
}
uint64_t orc_mul_modP(uint64_t x, uint64_t y) {          /* ModP.h:249-289 */
    return fold128((u128)x * y);
}
uint64_t orc_pow_modP(uint64_t x, uint64_t e) {
    uint64_t r = 1; x %= ORC_P;
    while (e) { if (e & 1) r = orc_mul_modP(r, x); x = orc_mul_modP(x, x); e >>= 1; }
    return r;
}
uint64_t orc_ls_modP(uint64_t x, int l) {                /* ModP.h:151-229: x*2^l */
    /* 2 has order 192 mod P (2^96 = -1) */
    return orc_mul_modP(x % ORC_P, orc_pow_modP(2, (uint64_t)(l % 192)));
}

/* ------------------------------------------------------------------------ */
/* transforms                                                                */
/* ------------------------------------------------------------------------ */
static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

static uint64_t root_of_len(int len) {                   /* Base.cu:63-70 */
    return orc_pow_modP(ORC_G, (uint64_t)(65536 / len));
}

void orc_ntt_naive(uint64_t *dst, const uint32_t *src, int len) {
    /* tests/test_ntt.cu:44-55 verbatim semantics */
    uint64_t w0 = root_of_len(len);
    uint64_t *r = (uint64_t *)malloc(sizeof(uint64_t) * len);
    r[0] = 1;
    for (int i = 1; i < len; i++) r[i] = orc_mul_modP(r[i - 1], w0);
    for (int i = 0; i < len; i++) {
        uint64_t acc = 0;
        for (int j = 0; j < len / 2; j++)
            acc = orc_add_modP(acc, orc_mul_modP(src[j], r[((long long)i * j) % len]));
        dst[i] = acc;
    }
    free(r);
}

/* in-place iterative radix-2 DIT, natural order in and out */
static void fft_inplace(uint64_t *a, int len, uint64_t w) {
    int lg = ilog2(len);
    for (int i = 0; i < len; i++) {
        int j = 0;
        for (int b = 0; b < lg; b++) if (i >> b & 1) j |= 1 << (lg - 1 - b);
        if (j > i) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    uint64_t *tw = (uint64_t *)malloc(sizeof(uint64_t) * (len / 2 > 0 ? len / 2 : 1));
    tw[0] = 1;
    for (int i = 1; i < len / 2; i++) tw[i] = orc_mul_modP(tw[i - 1], w);
    for (int h = 1; h < len; h <<= 1) {
        int step = len / (2 * h);
        for (int s = 0; s < len; s += 2 * h)
            for (int k = 0; k < h; k++) {
                uint64_t u = a[s + k], v = orc_mul_modP(a[s + k + h], tw[k * step]);
                a[s + k] = orc_add_modP(u, v);
                a[s + k + h] = orc_sub_modP(u, v);
            }
    }
    free(tw);
}

void orc_ntt_full(uint64_t *dst, const uint64_t *src, int len) {
    if (dst != src) memcpy(dst, src, sizeof(uint64_t) * len);
    for (int i = 0; i < len; i++) dst[i] %= ORC_P;
    fft_inplace(dst, len, root_of_len(len));
}

void orc_ntt_ext(uint64_t *dst, const uint32_t *src, int len) {
    /* zero-padded upper half: Base.cu:309 (ntt_1_*_ext loads only L/2 inputs) */
    for (int i = 0; i < len / 2; i++) dst[i] = src[i];
    for (int i = len / 2; i < len; i++) dst[i] = 0;
    fft_inplace(dst, len, root_of_len(len));
}

/* The transform as a host library would run it for throughput (bench.py cpu_baseline, round 6): the same radix-2 butterflies and
 * fold reduction as fft_inplace, with the twiddle and bit-reversal tables made ONCE per length and shared by the whole batch
 * (fft_inplace rebuilds both per transform), reduced operands throughout (one conditional subtraction per add / sub), OpenMP
 * over transforms.  Same outputs as orc_ntt_ext (tests/test_oracle_golden.py). */
typedef struct { int len; uint64_t w; uint64_t *tw; uint32_t *rev; } fast_tab;
static fast_tab g_fast[16];
static const fast_tab *fast_table_w(int len, uint64_t w) {          /* (made by the calling thread before any parallel region uses it) */
    fast_tab *t = NULL;
    for (int i = 0; i < 16; i++) if (g_fast[i].len == len && g_fast[i].w == w) return &g_fast[i];
    for (int i = 0; i < 16; i++) if (!g_fast[i].len) { t = &g_fast[i]; break; }
    if (!t) return NULL;
    const int lg = ilog2(len);
    t->tw = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(len / 2));
    t->rev = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)len);
    t->w = w;
    t->tw[0] = 1;
    for (int i = 1; i < len / 2; i++) t->tw[i] = orc_mul_modP(t->tw[i - 1], w);
    for (int i = 0; i < len; i++) { uint32_t j = 0; for (int b = 0; b < lg; b++) if (i >> b & 1) j |= 1u << (lg - 1 - b); t->rev[i] = j; }
    t->len = len;
    return t;
}
static const fast_tab *fast_table(int len) { return fast_table_w(len, root_of_len(len)); }
static inline uint64_t addp(uint64_t x, uint64_t y) { const uint64_t s = x + y; return (s < x || s >= ORC_P) ? s - ORC_P : s; }   /* x, y < P */
static inline uint64_t subp(uint64_t x, uint64_t y) { return x >= y ? x - y : x + (ORC_P - y); }
static void ntt_ext_fast(uint64_t *a, const uint32_t *src, int len, const fast_tab *T) {
    /* zero-padded input in bit-reversed order: sample j lands at rev[j]; the upper half of the input is zero (Base.cu:309) */
    memset(a, 0, sizeof(uint64_t) * (size_t)len);
    for (int j = 0; j < len / 2; j++) a[T->rev[j]] = src[j];
    for (int h = 1; h < len; h <<= 1) {
        const int step = len / (2 * h);
        for (int s = 0; s < len; s += 2 * h)
            for (int k = 0; k < h; k++) {
                const uint64_t u = a[s + k], v = fold128((u128)a[s + k + h] * T->tw[k * step]);
                a[s + k] = addp(u, v);
                a[s + k + h] = subp(u, v);
            }
    }
}
int orc_ntt_ext_fast_batch(uint64_t *dst, const uint32_t *src, int len, int batch, int threads) {
    const fast_tab *T = fast_table(len);
    if (!T) return -1;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < batch; b++)
        ntt_ext_fast(dst + (size_t)b * len, src + (size_t)b * (len / 2), len, T);
    return used;
}

/* batch of independent transforms on `threads` host threads (bench.py cpu_baseline leg) */
int orc_ntt_ext_batch(uint64_t *dst, const uint32_t *src, int len, int batch, int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < batch; b++)
        orc_ntt_ext(dst + (size_t)b * len, src + (size_t)b * (len / 2), len);
    return used;
}

uint64_t orc_len_inv(int len) { return orc_pow_modP((uint64_t)len, ORC_P - 2); }

void orc_intt_modp(uint32_t *dst, const uint64_t *src, int len, uint32_t p) {
    /* Base.cu:454,622,799: forward kernels on src[(L-idx)%L]; Base.cu:489: *L^-1 then %p */
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * len);
    for (int i = 0; i < len; i++) t[i] = src[(len - i) % len] % ORC_P;
    fft_inplace(t, len, root_of_len(len));
    uint64_t li = orc_len_inv(len);
    for (int i = 0; i < len; i++) dst[i] = (uint32_t)(orc_mul_modP(t[i], li) % p);
    free(t);
}

/* ------------------------------------------------------------------------ */
/* parameters                          (cuhe/Parameters.cu:53-145)           */
/* ------------------------------------------------------------------------ */
static int numbits_u64(uint64_t x) { int n = 0; while (x) { n++; x >>= 1; } return n; }
static uint64_t isqrt_u64(uint64_t x) {
    uint64_t r = 0, bit = 1ULL << 62;
    while (bit > x) bit >>= 2;
    while (bit) { if (x >= r + bit) { x -= r + bit; r = (r >> 1) + bit; } else r >>= 1; bit >>= 2; }
    return r;
}
static long euler_totient(long x) {                      /* Parameters.cu:35-52 */
    if (x < 3) return x;
    long res = x, n = x;
    for (long t = 2; t * t <= n; t++)
        if (n % t == 0) { while (n % t == 0) n /= t; res = res / t * (t - 1); }
    if (n > 1) res = res / n * (n - 1);
    return res;
}

int orc_set_param(orc_params *q, int d, int p, int w, int min, int cut, int m) {
    memset(q, 0, sizeof *q);
    q->depth = d; q->modMsg = p; q->logRelin = w;
    q->logCoeffMin = min; q->logCoeffCut = cut; q->mSize = m;
    q->logCoeffMax = min + cut * (d - 1);
    q->modLen = (int)euler_totient(m);
    q->modLen2 = 1 << numbits_u64((uint64_t)q->modLen - 1);
    if (q->modLen2 < 8192) q->modLen2 = 8192;
    q->rawLen = q->modLen2; q->crtLen = q->modLen2; q->nttLen = 2 * q->modLen2;
    q->logMsg = numbits_u64((uint64_t)p - 1);
    q->wordsMsg = (q->logMsg + 31) / 32;
    q->numEvalKey = w ? (q->logCoeffMax + w - 1) / w : 0;
    q->logCrtPrime = numbits_u64(isqrt_u64(ORC_P / (uint64_t)q->modLen));
    if (m == 131072) {
        /* NOT in the reference (its transforms stop at 65536 points, i.e. ring degree 2^15: Parameters.cu:63-68,
         * Base.cu:59-62): x^65536 + 1 through 64K-point negacyclic transforms.  nttLen is the transform length and the
         * primes obey the centred-lift bound 2 n p^2 < P: at most 23 bits. */
        q->nttLen = q->modLen2;
        q->logCrtPrime = numbits_u64(isqrt_u64(ORC_P / (2 * (uint64_t)q->modLen))) - 1;
    }
    q->numCrtPrime = (min + q->logCrtPrime - 1) / q->logCrtPrime;
    q->logCrtPrime = 0;
    while (q->logCrtPrime * q->numCrtPrime < min) q->logCrtPrime++;
    q->numCrtPrime += d - 1;
    return 0;
}
int orc_num_crt_prime(const orc_params *q, int lvl) {    /* Parameters.cu:107-116 */
    if (lvl == -1) return 1;
    return q->numCrtPrime - lvl;
}
int orc_log_coeff(const orc_params *q, int lvl) {        /* Parameters.cu:117-128 */
    if (lvl == -1) return q->logMsg;
    if (lvl < q->depth) return q->logCoeffMax - lvl * q->logCoeffCut;
    return q->logCoeffMin - q->logCrtPrime;
}
int orc_words_coeff(const orc_params *q, int lvl) {      /* Parameters.cu:129-132 */
    int t = (orc_log_coeff(q, lvl) + 31) / 32;
    return t > 1 ? t : 1;
}
int orc_num_eval_key(const orc_params *q, int lvl) {     /* Parameters.cu:133-135 */
    return (orc_log_coeff(q, lvl) + q->logRelin - 1) / q->logRelin;
}
int orc_get_level(const orc_params *q, int logq) {       /* Parameters.cu:136-141 */
    if (logq >= q->logCoeffMin) return (q->logCoeffMax - logq) / q->logCoeffCut;
    return -1;
}

/* ------------------------------------------------------------------------ */
/* primes                               (cuhe/Operations.cu:37-80)           */
/* ------------------------------------------------------------------------ */
static uint32_t powmod32(uint32_t b, uint32_t e, uint32_t m) {
    uint64_t r = 1, x = b % m;
    while (e) { if (e & 1) r = r * x % m; x = x * x % m; e >>= 1; }
    return (uint32_t)r;
}
int orc_is_prime_u32(uint32_t n) {
    if (n < 2) return 0;
    static const uint32_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (unsigned i = 0; i < sizeof small / sizeof *small; i++) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    uint32_t d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    static const uint32_t bases[] = {2, 7, 61};       /* deterministic below 2^32 */
    for (int bi = 0; bi < 3; bi++) {
        uint32_t a = bases[bi] % n;
        if (!a) continue;
        uint64_t x = powmod32(a, d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) { x = x * x % n; if (x == n - 1) { comp = 0; break; } }
        if (comp) return 0;
    }
    return 1;
}

int orc_gen_crt_primes(const orc_params *q, uint32_t *h_p) {
    int pnum = q->numCrtPrime, d = q->depth, l = q->logCrtPrime;
    int logmid = q->logCoeffMin - (pnum - d) * l;
    uint32_t temp = (uint32_t)((1u << l) - 1);
    for (int i = 0; i <= pnum - d - 1; i++) {             /* Operations.cu:44-50 */
        while (!orc_is_prime_u32(temp)) temp--;
        h_p[i] = temp; temp--;
    }
    uint32_t tmid = (logmid != l) ? (uint32_t)((1u << logmid) - 1) : temp;  /* :52-56 */
    while (!orc_is_prime_u32(tmid)) tmid--;
    h_p[pnum - d] = tmid;
    if (q->logCoeffCut == logmid) temp = tmid - 1;         /* :61-66 */
    else if (q->logCoeffCut == l) temp--;
    else temp = (uint32_t)((1u << q->logCoeffCut) - 1);
    for (int i = pnum - d + 1; i < pnum; i++) {            /* :67-73 */
        while (!orc_is_prime_u32(temp) || temp % (uint32_t)q->modMsg != 1) temp--;
        h_p[i] = temp; temp--;
    }
    return pnum;
}

/* ------------------------------------------------------------------------ */
/* cyclotomic polynomial                                                     */
/* ------------------------------------------------------------------------ */
static int mobius(int n) {
    int mu = 1;
    for (int p = 2; p * p <= n; p++)
        if (n % p == 0) { n /= p; if (n % p == 0) return 0; mu = -mu; }
    if (n > 1) mu = -mu;
    return mu;
}
int orc_cyclotomic(int m, int32_t *out, int cap) {
    /* Phi_m = prod_{d|m} (x^d - 1)^{mu(m/d)}; multiply first, then exact divisions */
    int deg = (int)euler_totient(m);
    if (m == 1) deg = 1;
    if (m == 2) deg = 1;
    long long *a = (long long *)calloc((size_t)2 * m + 2, sizeof(long long));
    int len = 1; a[0] = 1;
    for (int d = 1; d <= m; d++) {
        if (m % d || mobius(m / d) != 1) continue;
        /* a *= (x^d - 1) */
        for (int i = len - 1; i >= 0; i--) { a[i + d] += a[i]; a[i] = -a[i]; }
        len += d;
    }
    for (int d = 1; d <= m; d++) {
        if (m % d || mobius(m / d) != -1) continue;
        /* a /= (x^d - 1) exactly: a[i] = q[i-d] - q[i]  =>  q[i] = q[i-d] - a[i] */
        for (int i = 0; i < len - d; i++)
            a[i] = (i >= d ? a[i - d] : 0) - a[i];
        for (int i = len - d; i < len; i++) a[i] = 0;
        len -= d;
    }
    if (len - 1 != deg || len > cap) { free(a); return -1; }
    for (int i = 0; i < len; i++) out[i] = (int32_t)a[i];
    free(a);
    return deg;
}

/* ------------------------------------------------------------------------ */
/* small big-unsigned helpers: LE u32 words                                  */
/* ------------------------------------------------------------------------ */
static void big_mul_u32(uint32_t *a, int *len, uint32_t m) {
    uint64_t c = 0;
    for (int i = 0; i < *len; i++) { uint64_t t = (uint64_t)a[i] * m + c; a[i] = (uint32_t)t; c = t >> 32; }
    if (c) a[(*len)++] = (uint32_t)c;
}
static uint32_t big_divrem_u32(uint32_t *q, const uint32_t *a, int len, uint32_t d) {
    uint64_t r = 0;
    for (int i = len - 1; i >= 0; i--) { uint64_t t = (r << 32) | a[i]; q[i] = (uint32_t)(t / d); r = t % d; }
    return (uint32_t)r;
}
static uint32_t big_mod_u32(const uint32_t *a, int len, uint32_t d) {
    uint64_t r = 0;
    for (int i = len - 1; i >= 0; i--) r = ((r << 32) | a[i]) % d;
    return (uint32_t)r;
}
static int big_cmp(const uint32_t *a, const uint32_t *b, int len) {
    for (int i = len - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
    return 0;
}
static uint32_t invmod32(uint32_t a, uint32_t m) {
    long long t = 0, nt = 1, r = m, nr = a % m;
    while (nr) { long long qq = r / nr, x; x = t - qq * nt; t = nt; nt = x; x = r - qq * nr; r = nr; nr = x; }
    if (t < 0) t += m;
    return (uint32_t)t;
}

/* ------------------------------------------------------------------------ */
/* context                                                                   */
/* ------------------------------------------------------------------------ */
struct orc_ctx {
    orc_params prm;
    uint32_t primes[ORC_MAX_PRIMES];
    uint32_t *invp;                     /* [i*(i-1)/2+j] = p_i^-1 mod p_j */
    /* per level: M (W words), mi[np][W], bi[np] */
    int depth;
    uint32_t **M, **mi, **bi;
    int32_t *modulus;                   /* modLen+1 coefficients, monic */
    int cyclo;                          /* modulus divides x^mSize - 1 */
    /* nonzero low terms of the modulus */
    int nnz; int *nz_idx; int32_t *nz_val;
    /* per prime Barrett tables: NTT(u mod p_i), NTT(m - x^n mod p_i), CRT(m mod p_i) */
    uint64_t *u_ntt, *m_ntt; uint32_t *m_crt;
};

const orc_params *orc_ctx_params(const orc_ctx *c) { return &c->prm; }
const uint32_t *orc_ctx_primes(const orc_ctx *c) { return c->primes; }
const uint32_t *orc_ctx_invp(const orc_ctx *c) { return c->invp; }

static uint32_t smod(int64_t v, uint32_t p) { int64_t r = v % (int64_t)p; return (uint32_t)(r < 0 ? r + p : r); }

orc_ctx *orc_ctx_create(int d, int p, int w, int min, int cut, int m, const int32_t *modulus) {
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    orc_set_param(&c->prm, d, p, w, min, cut, m);
    const orc_params *q = &c->prm;
    if (q->numCrtPrime > ORC_MAX_PRIMES) { free(c); return NULL; }
    orc_gen_crt_primes(q, c->primes);
    int pnum = q->numCrtPrime, n = q->modLen, L = q->nttLen;
    /* Operations.cu:91-99 genCrtInvPrimes */
    c->invp = (uint32_t *)calloc((size_t)pnum * (pnum - 1) / 2 + 1, sizeof(uint32_t));
    for (int i = 1; i < pnum; i++)
        for (int j = 0; j < i; j++)
            c->invp[i * (i - 1) / 2 + j] = invmod32(c->primes[i] % c->primes[j], c->primes[j]);
    /* Operations.cu:81-90,107-134: per-level M, m_i = M/p_i, b_i = m_i^-1 mod p_i */
    c->depth = d;
    c->M = (uint32_t **)calloc(d, sizeof(void *));
    c->mi = (uint32_t **)calloc(d, sizeof(void *));
    c->bi = (uint32_t **)calloc(d, sizeof(void *));
    for (int lvl = 0; lvl < d; lvl++) {
        int np = pnum - lvl;
        uint32_t *M = (uint32_t *)calloc(ORC_MAX_WORDS, sizeof(uint32_t));
        int len = 1; M[0] = 1;
        for (int j = 0; j < np; j++) big_mul_u32(M, &len, c->primes[j]);
        c->M[lvl] = M;
        c->mi[lvl] = (uint32_t *)calloc((size_t)np * ORC_MAX_WORDS, sizeof(uint32_t));
        c->bi[lvl] = (uint32_t *)calloc(np, sizeof(uint32_t));
        for (int i = 0; i < np; i++) {
            uint32_t *mi = c->mi[lvl] + (size_t)i * ORC_MAX_WORDS;
            big_divrem_u32(mi, M, ORC_MAX_WORDS, c->primes[i]);
            c->bi[lvl][i] = invmod32(big_mod_u32(mi, ORC_MAX_WORDS, c->primes[i]), c->primes[i]);
        }
    }
    /* modulus */
    c->modulus = (int32_t *)calloc(n + 1, sizeof(int32_t));
    if (modulus) { memcpy(c->modulus, modulus, sizeof(int32_t) * (n + 1)); c->cyclo = 0; }
    else {
        if (orc_cyclotomic(m, c->modulus, n + 1) != n) { fprintf(stderr, "oracle: cyclotomic failed\n"); abort(); }
        c->cyclo = 1;
    }
    c->nz_idx = (int *)malloc(sizeof(int) * n);
    c->nz_val = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int i = 0; i < n; i++) if (c->modulus[i]) { c->nz_idx[c->nnz] = i; c->nz_val[c->nnz++] = c->modulus[i]; }
    /* Operations.cu:213-238 setPolyModulus: u = x^(2n-1) div m ; m' = m - x^n ; per prime */
    c->u_ntt = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)pnum * L);
    c->m_ntt = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)pnum * L);
    c->m_crt = (uint32_t *)calloc((size_t)pnum * q->crtLen, sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)calloc(L, sizeof(uint32_t));
    uint32_t *rem = (uint32_t *)calloc((size_t)2 * n, sizeof(uint32_t));
    for (int i = 0; i < pnum; i++) {
        uint32_t pi = c->primes[i];
        /* long division of x^(2n-1) by m over Z_p: quotient has n coefficients */
        memset(rem, 0, sizeof(uint32_t) * 2 * n);
        rem[2 * n - 1] = 1;
        memset(tmp, 0, sizeof(uint32_t) * L);
        for (int k = 2 * n - 1; k >= n; k--) {
            uint32_t cf = rem[k];
            tmp[k - n] = cf;
            if (!cf) continue;
            for (int t = 0; t < c->nnz; t++) {
                int idx = k - n + c->nz_idx[t];
                uint32_t sub = (uint32_t)((uint64_t)cf * smod(c->nz_val[t], pi) % pi);
                rem[idx] = rem[idx] >= sub ? rem[idx] - sub : rem[idx] + pi - sub;
            }
        }
        orc_ntt_ext(c->u_ntt + (size_t)i * L, tmp, L);
        memset(tmp, 0, sizeof(uint32_t) * L);
        for (int k = 0; k < n; k++) tmp[k] = smod(c->modulus[k], pi);
        memcpy(c->m_crt + (size_t)i * q->crtLen, tmp, sizeof(uint32_t) * n);
        orc_ntt_ext(c->m_ntt + (size_t)i * L, tmp, L);
    }
    free(tmp); free(rem);
    return c;
}

void orc_ctx_destroy(orc_ctx *c) {
    if (!c) return;
    for (int l = 0; l < c->depth; l++) { free(c->M[l]); free(c->mi[l]); free(c->bi[l]); }
    free(c->M); free(c->mi); free(c->bi); free(c->invp); free(c->modulus);
    free(c->nz_idx); free(c->nz_val); free(c->u_ntt); free(c->m_ntt); free(c->m_crt);
    free(c);
}

int orc_ctx_coeff_modulus(const orc_ctx *c, int lvl, uint32_t *words, int cap) {
    int W = ORC_MAX_WORDS;
    while (W > 1 && c->M[lvl][W - 1] == 0) W--;
    if (W > cap) return -1;
    memcpy(words, c->M[lvl], sizeof(uint32_t) * W);
    return W;
}

/* ------------------------------------------------------------------------ */
/* stages                                                                    */
/* ------------------------------------------------------------------------ */
void orc_crt(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int lvl) {
    /* Base.cu:857-879: Horner from the top word; only idx < modLen written */
    const orc_params *q = &c->prm;
    int np = orc_num_crt_prime(q, lvl), W = orc_words_coeff(q, lvl);
    memset(dst, 0, sizeof(uint32_t) * (size_t)np * q->crtLen);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < np; i++) {
        uint32_t p = c->primes[i];
        for (int idx = 0; idx < q->modLen; idx++) {
            const uint32_t *co = src + (size_t)idx * W;
            uint32_t l = co[W - 1] % p;
            for (int k = W - 2; k >= 0; k--)
                l = (uint32_t)((((uint64_t)l << 32) + (co[k] % p)) % p);
            dst[(size_t)i * q->crtLen + idx] = l;
        }
    }
}

void orc_icrt(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int lvl) {
    /* Base.cu:880-924: sum_i ((x_i*b_i) mod p_i) * m_i, conditional -M after each term */
    const orc_params *q = &c->prm;
    int np = orc_num_crt_prime(q, lvl), W = orc_words_coeff(q, lvl);
    const uint32_t *M = c->M[lvl];
    memset(dst, 0, sizeof(uint32_t) * (size_t)q->rawLen * W);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int idx = 0; idx < q->modLen; idx++) {
        uint32_t acc[ORC_MAX_WORDS + 2];
        memset(acc, 0, sizeof acc);
        for (int i = 0; i < np; i++) {
            uint32_t p = c->primes[i];
            uint64_t tar = src[(size_t)i * q->crtLen + idx] % p;
            uint32_t tt = (uint32_t)(tar * c->bi[lvl][i] % p);
            const uint32_t *mi = c->mi[lvl] + (size_t)i * ORC_MAX_WORDS;
            uint64_t carry = 0;
            for (int k = 0; k <= W; k++) {
                uint64_t t = (uint64_t)tt * (k < ORC_MAX_WORDS ? mi[k] : 0) + acc[k] + carry;
                acc[k] = (uint32_t)t; carry = t >> 32;
            }
            if (big_cmp(acc, M, W + 1) >= 0) {
                uint64_t br = 0;
                for (int k = 0; k <= W; k++) {
                    uint64_t t = (uint64_t)acc[k] - M[k] - br;
                    acc[k] = (uint32_t)t; br = (t >> 32) & 1;
                }
            }
        }
        memcpy(dst + (size_t)idx * W, acc, sizeof(uint32_t) * W);
    }
}

void orc_ntt(const orc_ctx *c, uint64_t *dst, const uint32_t *src, int np) {
    const orc_params *q = &c->prm;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < np; i++)
        orc_ntt_ext(dst + (size_t)i * q->nttLen, src + (size_t)i * q->crtLen, q->nttLen);
}
void orc_intt_hold(const orc_ctx *c, uint32_t *dst, const uint64_t *src, int np) {
    const orc_params *q = &c->prm;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < np; i++)
        orc_intt_modp(dst + (size_t)i * q->nttLen, src + (size_t)i * q->nttLen, q->nttLen, c->primes[i]);
}
void orc_intt(const orc_ctx *c, uint32_t *dst, const uint64_t *src, int np) {
    const orc_params *q = &c->prm;
    uint32_t *t = (uint32_t *)malloc(sizeof(uint32_t) * q->nttLen);
    for (int i = 0; i < np; i++) {
        orc_intt_modp(t, src + (size_t)i * q->nttLen, q->nttLen, c->primes[i]);
        memcpy(dst + (size_t)i * q->crtLen, t, sizeof(uint32_t) * q->crtLen);
    }
    free(t);
}

void orc_poly_reduce_exact(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int np) {
    const orc_params *q = &c->prm;
    int n = q->modLen, L = q->nttLen, m = q->mSize;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int i = 0; i < np; i++) {
        uint32_t *f = (uint32_t *)malloc(sizeof(uint32_t) * L);
        uint32_t p = c->primes[i];
        for (int k = 0; k < L; k++) f[k] = src[(size_t)i * L + k] % p;
        int top = L - 1;
        if (c->cyclo && m > n && m < L) {          /* Phi_m | x^m - 1: fold first */
            for (int k = L - 1; k >= m; k--) {
                uint32_t s = f[k - m] + f[k]; f[k - m] = s >= p ? s - p : s; f[k] = 0;
            }
            top = m - 1;
        }
        for (int k = top; k >= n; k--) {
            uint32_t cf = f[k];
            if (!cf) continue;
            f[k] = 0;
            for (int t = 0; t < c->nnz; t++) {
                int idx = k - n + c->nz_idx[t];
                uint32_t sub = (uint32_t)((uint64_t)cf * smod(c->nz_val[t], p) % p);
                f[idx] = f[idx] >= sub ? f[idx] - sub : f[idx] + p - sub;
            }
        }
        memset(dst + (size_t)i * q->crtLen, 0, sizeof(uint32_t) * q->crtLen);
        memcpy(dst + (size_t)i * q->crtLen, f, sizeof(uint32_t) * n);
        free(f);
    }
}

void orc_barrett(const orc_ctx *c, uint32_t *dst, const uint32_t *src_in, int np) {
    /* Operations.cu:460-501, one prime at a time */
    const orc_params *q = &c->prm;
    int n = q->modLen, L = q->nttLen, cl = q->crtLen;
    uint32_t *src = (uint32_t *)malloc(sizeof(uint32_t) * L);
    uint32_t *crt = (uint32_t *)malloc(sizeof(uint32_t) * L);
    uint64_t *ntt = (uint64_t *)malloc(sizeof(uint64_t) * L);
    for (int i = 0; i < np; i++) {
        uint32_t p = c->primes[i];
        memcpy(src, src_in + (size_t)i * L, sizeof(uint32_t) * L);       /* :465-467 */
        orc_ntt_ext(ntt, src + n - 1, L);                                /* :469-470 */
        for (int k = 0; k < L; k++)                                      /* :472 barrett_mul_un */
            ntt[k] = orc_mul_modP(ntt[k], c->u_ntt[(size_t)i * L + k]);
        orc_intt_modp(crt, ntt, L, p);                                   /* :474 */
        memset(crt, 0, sizeof(uint32_t) * n);                            /* :476-478 */
        orc_ntt_ext(ntt, crt + n, L);                                    /* :480-481 */
        for (int k = 0; k < L; k++)                                      /* :483 barrett_mul_mn */
            ntt[k] = orc_mul_modP(ntt[k], c->m_ntt[(size_t)i * L + k]);
        for (int k = 0; k < n; k++) {                                    /* :486 barrett_sub_1 */
            uint32_t a = src[n + k], b = crt[n + k];
            if (a < b) a += p;
            src[n + k] = a - b;
        }
        orc_intt_modp(crt, ntt, L, p);                                   /* :489 */
        for (int k = 0; k < L; k++) {                                    /* :491 barrett_sub_2 */
            uint32_t a = src[k], b = crt[k];
            if (a < b) a += p;
            src[k] = a - b;
        }
        if (src[n] > 0)                                                  /* :494 barrett_sub_mc */
            for (int k = 0; k < n - 1; k++) {
                uint32_t dd = src[k], s = c->m_crt[(size_t)i * cl + k];
                if (dd < s) dd += p;
                src[k] = dd - s;
            }
        memcpy(dst + (size_t)i * cl, src, sizeof(uint32_t) * cl);        /* :498-500 */
    }
    free(src); free(crt); free(ntt);
}

void orc_intt_mod(const orc_ctx *c, uint32_t *dst, const uint64_t *src, int np) {
    const orc_params *q = &c->prm;
    uint32_t *hold = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np * q->nttLen);
    orc_intt_hold(c, hold, src, np);
    orc_poly_reduce_exact(c, dst, hold, np);
    free(hold);
}

void orc_ntt_mul(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *y, int np) {
    size_t n = (size_t)np * c->prm.nttLen;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (size_t i = 0; i < n; i++) z[i] = orc_mul_modP(x[i], y[i]);
}
void orc_ntt_add(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *y, int np) {
    size_t n = (size_t)np * c->prm.nttLen;
    for (size_t i = 0; i < n; i++) z[i] = orc_add_modP(x[i], y[i]);
}
void orc_ntt_mul_nx1(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *s, int np) {
    int L = c->prm.nttLen;
    for (int i = 0; i < np; i++) for (int k = 0; k < L; k++)
        z[(size_t)i * L + k] = orc_mul_modP(x[(size_t)i * L + k], s[k]);
}
void orc_ntt_add_nx1(const orc_ctx *c, uint64_t *z, const uint64_t *x, const uint64_t *s, int np) {
    int L = c->prm.nttLen;
    for (int i = 0; i < np; i++) for (int k = 0; k < L; k++)
        z[(size_t)i * L + k] = orc_add_modP(x[(size_t)i * L + k], s[k]);
}

void orc_crt_add(const orc_ctx *c, uint32_t *z, const uint32_t *x, const uint32_t *y, int np) {
    const orc_params *q = &c->prm;                         /* Base.cu:1088-1095 */
    for (int i = 0; i < np; i++) for (int k = 0; k < q->modLen; k++) {
        size_t o = (size_t)i * q->crtLen + k;
        z[o] = (x[o] + y[o]) % c->primes[i];
    }
}
void orc_crt_add_int(const orc_ctx *c, uint32_t *z, const uint32_t *x, unsigned a, int np) {
    const orc_params *q = &c->prm;                         /* Base.cu:1096-1100: constant term only */
    for (int i = 0; i < np; i++) {
        size_t o = (size_t)i * q->crtLen;
        z[o] = (x[o] + (a % c->primes[i])) % c->primes[i];
    }
}
void orc_crt_add_nx1(const orc_ctx *c, uint32_t *z, const uint32_t *x, const uint32_t *s, int np) {
    const orc_params *q = &c->prm;                         /* Base.cu:1101-1109 */
    for (int i = 0; i < np; i++) for (int k = 0; k < q->modLen; k++) {
        size_t o = (size_t)i * q->crtLen + k;
        z[o] = (x[o] + s[k]) % c->primes[i];
    }
}

void orc_modswitch(const orc_ctx *c, uint32_t *dst, const uint32_t *src, int np) {
    /* Base.cu:1112-1138 */
    const orc_params *q = &c->prm;
    int cl = q->crtLen, modmsg = q->modMsg;
    uint32_t pt = c->primes[np - 1];
    for (int idx = 0; idx < q->modLen; idx++) {
        int dirty = (int)src[(size_t)(np - 1) * cl + idx];
        int ep = dirty % modmsg;
        if (ep != 0) {
            if ((uint32_t)dirty > ((pt - 1) / 2)) dirty -= ep * (int)pt;
            else dirty += ep * (int)pt;
        }
        for (int i = 0; i < np - 1; i++) {
            int64_t temp = (int64_t)src[(size_t)i * cl + idx];
            while (temp < dirty) temp += c->primes[i];
            temp -= dirty;
            uint64_t tt = (uint64_t)temp;
            tt *= c->invp[(np - 1) * (np - 2) / 2 + i];
            tt %= c->primes[i];
            dst[(size_t)i * cl + idx] = (uint32_t)tt;
        }
    }
}

static uint32_t window_of(const uint32_t *co, int W, int w, int wid) {
    /* Base.cu:361-371 */
    int wi = (w * wid) >> 5;
    uint64_t s;
    if (wi + 1 < W) s = ((uint64_t)co[wi + 1] << 32) + co[wi];
    else s = co[wi];
    s >>= (w * wid) & 0x1f;
    s &= (uint64_t)((1u << w) - 1);
    return (uint32_t)s;
}

void orc_nttw(const orc_ctx *c, uint64_t *dst, const uint32_t *raw, int lvl) {
    const orc_params *q = &c->prm;
    int k = orc_num_eval_key(q, lvl), W = orc_words_coeff(q, lvl), L = q->nttLen;
    uint32_t *win = (uint32_t *)malloc(sizeof(uint32_t) * q->crtLen);
    for (int j = 0; j < k; j++) {
        for (int idx = 0; idx < q->crtLen; idx++)
            win[idx] = window_of(raw + (size_t)idx * W, W, q->logRelin, j);
        orc_ntt_ext(dst + (size_t)j * L, win, L);
    }
    free(win);
}

void orc_init_relin(const orc_ctx *c, uint64_t *ek, const uint32_t *evalkey_raw) {
    /* Relinearization.cu:43-57: ek[prime][key][nttLen] */
    const orc_params *q = &c->prm;
    int K = q->numEvalKey, np = q->numCrtPrime, W0 = orc_words_coeff(q, 0), L = q->nttLen;
    uint32_t *crt = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np * q->crtLen);
    for (int j = 0; j < K; j++) {
        orc_crt(c, crt, evalkey_raw + (size_t)j * q->rawLen * W0, 0);
        for (int i = 0; i < np; i++)
            orc_ntt_ext(ek + ((size_t)i * K + j) * L, crt + (size_t)i * q->crtLen, L);
    }
    free(crt);
}

void orc_relin(const orc_ctx *c, uint64_t *dst, const uint32_t *raw, int lvl, const uint64_t *ek) {
    /* Relinearization.cu:76-88 + Base.cu:1024-1033 */
    const orc_params *q = &c->prm;
    int k = orc_num_eval_key(q, lvl), np = orc_num_crt_prime(q, lvl), K = q->numEvalKey, L = q->nttLen;
    uint64_t *cw = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)k * L);
    orc_nttw(c, cw, raw, lvl);
    for (int i = 0; i < np; i++)
        for (int idx = 0; idx < L; idx++) {
            uint64_t sum = 0;
            for (int j = 0; j < k; j++)
                sum = orc_add_modP(sum, orc_mul_modP(cw[(size_t)j * L + idx], ek[((size_t)i * K + j) * L + idx]));
            dst[(size_t)i * L + idx] = sum;
        }
    free(cw);
}

void orc_mul_raw(const orc_ctx *c, uint32_t *out, const uint32_t *a, const uint32_t *b, int lvl) {
    /* CuHE.cu:259-268 */
    const orc_params *q = &c->prm;
    int np = orc_num_crt_prime(q, lvl);
    size_t cn = (size_t)np * q->crtLen, nn = (size_t)np * q->nttLen;
    uint32_t *ca = (uint32_t *)malloc(sizeof(uint32_t) * cn), *cb = (uint32_t *)malloc(sizeof(uint32_t) * cn);
    uint64_t *na = (uint64_t *)malloc(sizeof(uint64_t) * nn), *nb = (uint64_t *)malloc(sizeof(uint64_t) * nn);
    orc_crt(c, ca, a, lvl); orc_crt(c, cb, b, lvl);
    orc_ntt(c, na, ca, np); orc_ntt(c, nb, cb, np);
    orc_ntt_mul(c, na, na, nb, np);
    orc_intt_mod(c, ca, na, np);
    orc_icrt(c, out, ca, lvl);
    free(ca); free(cb); free(na); free(nb);
}

void orc_mul_relin_crt(const orc_ctx *c, uint32_t *dst, const uint32_t *a, const uint32_t *b,
                       int lvl, const uint64_t *ek) {
    /* cAnd (CuHE.cu:101) then CuCtxt::relin (CuHE.cu:570-581) */
    const orc_params *q = &c->prm;
    int np = orc_num_crt_prime(q, lvl), W = orc_words_coeff(q, lvl);
    size_t cn = (size_t)np * q->crtLen, nn = (size_t)np * q->nttLen;
    uint32_t *cr = (uint32_t *)malloc(sizeof(uint32_t) * cn);
    uint32_t *raw = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)q->rawLen * W);
    uint64_t *na = (uint64_t *)malloc(sizeof(uint64_t) * nn), *nb = (uint64_t *)malloc(sizeof(uint64_t) * nn);
    orc_ntt(c, na, a, np); orc_ntt(c, nb, b, np);
    orc_ntt_mul(c, na, na, nb, np);
    orc_intt_mod(c, cr, na, np);          /* x2r: n2c (isProd) */
    orc_icrt(c, raw, cr, lvl);            /*      c2r          */
    orc_relin(c, na, raw, lvl, ek);
    orc_intt_mod(c, dst, na, np);         /* n2c (isProd) */
    free(cr); free(raw); free(na); free(nb);
}

/* ------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------ */
/* products modulo x^n + 1 (m = 2n a power of two)                           */
/* The reference reduces the zero-padded cyclic product with its NTT Barrett */
/* chain (cuhe/Operations.cu:460-501); for Phi_m = x^n + 1 that remainder is */
/* the negacyclic convolution restated here, first by definition, then with  */
/* the twisted length-n transform the MI355X backend uses on such rings.     */
/* ------------------------------------------------------------------------ */
void orc_negacyclic_mul_modp_naive(uint32_t *dst, const uint32_t *a, const uint32_t *b, int n, uint32_t p) {
    /* c[i] = sum_{j<=i} a[j] b[i-j] - sum_{j>i} a[j] b[n+i-j]  (mod p), residues below p < 2^32 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        uint64_t pos = 0, neg = 0;
        for (int j = 0; j <= i; j++) pos = (pos + (uint64_t)a[j] * b[i - j] % p) % p;
        for (int j = i + 1; j < n; j++) neg = (neg + (uint64_t)a[j] * b[n + i - j] % p) % p;
        dst[i] = (uint32_t)((pos + p - neg) % p);
    }
}
/* psi = 7^((P-1)/2n): 7 generates Z_P^* (P - 1 = 2^32 * 3 * 5 * 17 * 257 * 65537), so psi is a primitive 2n-th root of
 * unity for every power of two 2n <= 2^32.  Which primitive root is used does not matter for the product. */
static uint64_t nc_psi(int n) {
    uint64_t psi = orc_pow_modP(7, (ORC_P - 1) / (2 * (uint64_t)n));
    if (orc_pow_modP(psi, (uint64_t)n) != ORC_P - 1) { fprintf(stderr, "oracle: 7 is not a generator?\n"); abort(); }
    return psi;
}
void orc_nc_ntt(uint64_t *dst, const uint32_t *src, int n) {     /* X[k] = sum_j x[j] psi^(j(2k+1)) */
    const uint64_t psi = nc_psi(n);
    uint64_t t = 1;
    for (int j = 0; j < n; j++) { dst[j] = orc_mul_modP(src[j], t); t = orc_mul_modP(t, psi); }
    fft_inplace(dst, n, orc_mul_modP(psi, psi));
}
int orc_nc_intt_modp(uint32_t *dst, const uint64_t *src, int n, uint32_t p) {
    /* exact integer coefficient = centred representative modulo P (|c| < P/2 required), then mod p */
    const uint64_t psi = nc_psi(n), ipsi = orc_pow_modP(psi, ORC_P - 2);
    const uint64_t w = orc_mul_modP(psi, psi), iw = orc_pow_modP(w, ORC_P - 2), ninv = orc_pow_modP((uint64_t)n, ORC_P - 2);
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * n);
    memcpy(t, src, sizeof(uint64_t) * n);
    fft_inplace(t, n, iw);
    uint64_t tw = ninv;
    for (int j = 0; j < n; j++) {
        const uint64_t v = orc_mul_modP(t[j], tw);
        tw = orc_mul_modP(tw, ipsi);
        if (v > ORC_P / 2) { const uint32_t r = (uint32_t)((ORC_P - v) % p); dst[j] = r ? p - r : 0; }
        else dst[j] = (uint32_t)(v % p);
    }
    free(t);
    return 0;
}
int orc_negacyclic_mul_modp(uint32_t *dst, const uint32_t *a, const uint32_t *b, int n, uint32_t p) {
    if ((u128)2 * n * (p - 1) * (p - 1) >= ORC_P) return -1;        /* the centred lift would be ambiguous */
    uint64_t *A = (uint64_t *)malloc(sizeof(uint64_t) * n), *B = (uint64_t *)malloc(sizeof(uint64_t) * n);
    orc_nc_ntt(A, a, n); orc_nc_ntt(B, b, n);
    for (int i = 0; i < n; i++) A[i] = orc_mul_modP(A[i], B[i]);
    orc_nc_intt_modp(dst, A, n, p);
    free(A); free(B);
    return 0;
}
/* key-switch inner product modulo x^n + 1 for ONE prime: dst = sum_j win[j] * key[j] mod (x^n + 1) mod p, accumulated in the
 * transform domain like cuhe/Relinearization.cu:76-88 (windows below 2^w, key residues below p; 2 k n 2^w p < P required) */
int orc_nc_relin_modp(uint32_t *dst, const uint32_t *win, const uint32_t *key, int k, int n, uint32_t p) {
    uint64_t *acc = (uint64_t *)calloc(n, sizeof(uint64_t)), *A = (uint64_t *)malloc(sizeof(uint64_t) * n), *B = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (int j = 0; j < k; j++) {
        orc_nc_ntt(A, win + (size_t)j * n, n); orc_nc_ntt(B, key + (size_t)j * n, n);
        for (int i = 0; i < n; i++) acc[i] = orc_add_modP(acc[i], orc_mul_modP(A[i], B[i]));
    }
    orc_nc_intt_modp(dst, acc, n, p);
    free(acc); free(A); free(B);
    return 0;
}

/* cAnd then CuCtxt::relin (cuhe/CuHE.cu:101,570-581; Relinearization.cu:76-88) for B ciphertext pairs on a ring x^n + 1 at the
 * sizes of BASELINE config 4, where orc_mul_relin_crt's key table (np x K x nttLen words) no longer fits: the same chain with
 * every product modulo x^n + 1 taken per prime through the negacyclic restatement above (pinned against the cyclic chain by
 * tests/test_oracle_negacyclic.py), the key transforms formed once per prime and shared by the B pairs.
 *   a, b, dst: u32[B][np][crtLen] reduced CRT rows of level lvl; ekc: u32[K][np0][crtLen] = orc_crt of the K raw keys (level 0).
 * Returns -1 when a centred lift would be ambiguous (2 n p^2 >= P or 2 k n 2^w p >= P). */
int orc_nc_mul_relin_crt_batch(const orc_ctx *c, uint32_t *dst, const uint32_t *a, const uint32_t *b, int B, int lvl, const uint32_t *ekc) {
    const orc_params *q = &c->prm;
    const int n = q->modLen, cl = q->crtLen, np = orc_num_crt_prime(q, lvl), np0 = q->numCrtPrime, W = orc_words_coeff(q, lvl);
    const int k = orc_num_eval_key(q, lvl), w = q->logRelin;
    if (n != cl || (n & (n - 1))) return -1;
    for (int i = 0; i < np; i++) {
        const u128 p = c->primes[i];
        if ((u128)2 * n * (p - 1) * (p - 1) >= ORC_P || (u128)2 * k * n * (((u128)1 << w) - 1) * (p - 1) >= ORC_P) return -1;
    }
    const size_t cn = (size_t)np * cl;
    uint32_t *cr = (uint32_t *)malloc(sizeof(uint32_t) * cn);
    uint32_t *raw = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)q->rawLen * W);
    uint32_t *win = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *wn = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)B * k * n);          /* transforms of the windows of every pair */
    for (int t = 0; t < B; t++) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
        for (int i = 0; i < np; i++)                                                   /* cAnd ; n2c (isProd) */
            orc_negacyclic_mul_modp(cr + (size_t)i * cl, a + t * cn + (size_t)i * cl, b + t * cn + (size_t)i * cl, n, c->primes[i]);
        orc_icrt(c, raw, cr, lvl);                                                     /* c2r */
        for (int j = 0; j < k; j++) {                                                  /* Base.cu:345-385: windows, then their transforms */
            for (int idx = 0; idx < n; idx++) win[idx] = window_of(raw + (size_t)idx * W, W, w, j);
            orc_nc_ntt(wn + ((size_t)t * k + j) * n, win, n);
        }
    }
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1)
    for (int i = 0; i < np; i++) {                                                     /* Base.cu:1024-1033 per prime, then n2c (isProd) */
        uint64_t *acc = (uint64_t *)calloc((size_t)B * n, sizeof(uint64_t)), *key = (uint64_t *)malloc(sizeof(uint64_t) * n);
        for (int j = 0; j < k; j++) {
            orc_nc_ntt(key, ekc + ((size_t)j * np0 + i) * cl, n);
            for (int t = 0; t < B; t++) {
                const uint64_t *x = wn + ((size_t)t * k + j) * n;
                uint64_t *y = acc + (size_t)t * n;
                for (int idx = 0; idx < n; idx++) y[idx] = orc_add_modP(y[idx], orc_mul_modP(x[idx], key[idx]));
            }
        }
        for (int t = 0; t < B; t++) orc_nc_intt_modp(dst + t * cn + (size_t)i * cl, acc + (size_t)t * n, n, c->primes[i]);
        free(acc); free(key);
    }
    free(cr); free(raw); free(win); free(wn);
    return 0;
}

/* ---- bench.py's CPU leg for "ciphertext mul + relin" (round 6): the chain above for ONE pair with the evaluation keys transformed
 * beforehand (the GPU library keeps them resident in transformed form: the timed region of both sides starts from the same state),
 * the table-sharing transforms, OpenMP over primes / windows / coefficients.  Same rows as orc_nc_mul_relin_crt_batch with B = 1
 * (tests/test_oracle_negacyclic.py). */
struct orc_nc_prepared {
    const orc_ctx *c; int lvl, n, np, k;
    uint64_t *keys;                /* u64[np][k][n]: negacyclic transforms of the keys' CRT rows */
    uint64_t *tw, *itw;            /* psi^j ; n^-1 psi^-j */
    const fast_tab *F, *I;
};
static void nc_fft(uint64_t *a, const fast_tab *T) {
    const int len = T->len;
    for (int i = 0; i < len; i++) { const uint32_t j = T->rev[i]; if (j > (uint32_t)i) { const uint64_t t = a[i]; a[i] = a[j]; a[j] = t; } }
    for (int h = 1; h < len; h <<= 1) {
        const int step = len / (2 * h);
        for (int s = 0; s < len; s += 2 * h)
            for (int k = 0; k < h; k++) {
                const uint64_t u = a[s + k], v = fold128((u128)a[s + k + h] * T->tw[k * step]);
                a[s + k] = addp(u, v);
                a[s + k + h] = subp(u, v);
            }
    }
}
static void nc_fwd(const orc_nc_prepared *P, uint64_t *dst, const uint32_t *src) {
    for (int j = 0; j < P->n; j++) dst[j] = fold128((u128)src[j] * P->tw[j]);
    nc_fft(dst, P->F);
}
static void nc_inv_modp(const orc_nc_prepared *P, uint32_t *dst, uint64_t *t, uint32_t p) {     /* t is overwritten */
    nc_fft(t, P->I);
    for (int j = 0; j < P->n; j++) {
        const uint64_t v = fold128((u128)t[j] * P->itw[j]);
        if (v > ORC_P / 2) { const uint32_t r = (uint32_t)((ORC_P - v) % p); dst[j] = r ? p - r : 0; }
        else dst[j] = (uint32_t)(v % p);
    }
}
orc_nc_prepared *orc_nc_prepare(const orc_ctx *c, int lvl, const uint32_t *ekc) {
    const orc_params *q = &c->prm;
    const int n = q->modLen, cl = q->crtLen, np = orc_num_crt_prime(q, lvl), np0 = q->numCrtPrime, k = orc_num_eval_key(q, lvl), w = q->logRelin;
    if (n != cl || (n & (n - 1))) return NULL;
    for (int i = 0; i < np; i++) {
        const u128 p = c->primes[i];
        if ((u128)2 * n * (p - 1) * (p - 1) >= ORC_P || (u128)2 * k * n * (((u128)1 << w) - 1) * (p - 1) >= ORC_P) return NULL;
    }
    orc_nc_prepared *P = (orc_nc_prepared *)calloc(1, sizeof *P);
    P->c = c; P->lvl = lvl; P->n = n; P->np = np; P->k = k;
    const uint64_t psi = nc_psi(n), ipsi = orc_pow_modP(psi, ORC_P - 2), wr = orc_mul_modP(psi, psi);
    P->F = fast_table_w(n, wr); P->I = fast_table_w(n, orc_pow_modP(wr, ORC_P - 2));
    if (!P->F || !P->I) { free(P); return NULL; }
    P->tw = (uint64_t *)malloc(sizeof(uint64_t) * n); P->itw = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t t = 1, it = orc_pow_modP((uint64_t)n, ORC_P - 2);
    for (int j = 0; j < n; j++) { P->tw[j] = t; P->itw[j] = it; t = orc_mul_modP(t, psi); it = orc_mul_modP(it, ipsi); }
    P->keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)np * k * n);
    if (!P->keys) { free(P->tw); free(P->itw); free(P); return NULL; }
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1) collapse(2)
    for (int i = 0; i < np; i++)
        for (int j = 0; j < k; j++) nc_fwd(P, P->keys + ((size_t)i * k + j) * n, ekc + ((size_t)j * np0 + i) * cl);
    return P;
}
void orc_nc_prepared_free(orc_nc_prepared *P) { if (P) { free(P->keys); free(P->tw); free(P->itw); free(P); } }
int orc_nc_mul_relin_prepared(const orc_nc_prepared *P, uint32_t *dst, const uint32_t *a, const uint32_t *b) {
    const orc_ctx *c = P->c; const orc_params *q = &c->prm;
    const int n = P->n, np = P->np, k = P->k, W = orc_words_coeff(q, P->lvl), w = q->logRelin;
    uint32_t *cr = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)np * n);
    uint32_t *raw = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)q->rawLen * W);
    uint64_t *wn = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)k * n);
#pragma omp parallel num_threads(g_threads)
    {
        uint64_t *A = (uint64_t *)malloc(sizeof(uint64_t) * n), *B = (uint64_t *)malloc(sizeof(uint64_t) * n);
        uint32_t *win = (uint32_t *)malloc(sizeof(uint32_t) * n);
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < np; i++) {                                   /* cAnd ; n2c (isProd) */
            nc_fwd(P, A, a + (size_t)i * n); nc_fwd(P, B, b + (size_t)i * n);
            for (int x = 0; x < n; x++) A[x] = fold128((u128)A[x] * B[x]);
            nc_inv_modp(P, cr + (size_t)i * n, A, c->primes[i]);
        }
#pragma omp single
        orc_icrt(c, raw, cr, P->lvl);                                    /* c2r (its own parallel loop over coefficients) */
#pragma omp for schedule(dynamic, 1)
        for (int j = 0; j < k; j++) {                                    /* windows and their transforms (Base.cu:345-385) */
            for (int x = 0; x < n; x++) win[x] = window_of(raw + (size_t)x * W, W, w, j);
            nc_fwd(P, wn + (size_t)j * n, win);
        }
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < np; i++) {                                   /* Base.cu:1024-1033 per prime, then n2c (isProd) */
            memset(A, 0, sizeof(uint64_t) * n);
            for (int j = 0; j < k; j++) {
                const uint64_t *x = wn + (size_t)j * n, *y = P->keys + ((size_t)i * k + j) * n;
                for (int t = 0; t < n; t++) A[t] = addp(A[t], fold128((u128)x[t] * y[t]));
            }
            nc_inv_modp(P, dst + (size_t)i * n, A, c->primes[i]);
        }
        free(A); free(B); free(win);
    }
    free(cr); free(raw); free(wn);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* optional CPU baseline: the multiplication the reference delegates to NTL   */
/* (examples/DHS/DHS.cu:219-221: t = a * b; t %= Phi; coefficients mod q) done */
/* the way a big-number library does it -- Kronecker substitution, ONE mpz_mul */
/* of two integers of n * S bits -- with GMP opened at run time when the box   */
/* has it (NTL itself is absent from this image; NTL builds on GMP).  Used by  */
/* bench.py as a labelled second cpu_baseline and by a CPU test against the    */
/* oracle's own path; never by the product.                                    */
/* ------------------------------------------------------------------------ */
#include <dlfcn.h>
typedef struct { int alloc, size; unsigned long *d; } gmp_mpz[1];
static struct {
    void *h; int tried;
    void (*init)(gmp_mpz); void (*clear)(gmp_mpz);
    void (*import)(gmp_mpz, size_t, int, size_t, int, size_t, const void *);
    void *(*export)(void *, size_t *, int, size_t, int, size_t, const gmp_mpz);
    void (*mul)(gmp_mpz, const gmp_mpz, const gmp_mpz); void (*sub)(gmp_mpz, const gmp_mpz, const gmp_mpz);
    void (*fdiv_r)(gmp_mpz, const gmp_mpz, const gmp_mpz);
} GMP;
static int gmp_load(void) {
    if (GMP.tried) return GMP.h != NULL;
    GMP.tried = 1;
    const char *names[] = {"libgmp.so.10", "libgmp.so", "/opt/conda/lib/libgmp.so"};
    for (int i = 0; i < 3 && !GMP.h; i++) GMP.h = dlopen(names[i], RTLD_NOW);
    if (!GMP.h) return 0;
    *(void **)&GMP.init = dlsym(GMP.h, "__gmpz_init"); *(void **)&GMP.clear = dlsym(GMP.h, "__gmpz_clear");
    *(void **)&GMP.import = dlsym(GMP.h, "__gmpz_import"); *(void **)&GMP.export = dlsym(GMP.h, "__gmpz_export");
    *(void **)&GMP.mul = dlsym(GMP.h, "__gmpz_mul"); *(void **)&GMP.sub = dlsym(GMP.h, "__gmpz_sub");
    *(void **)&GMP.fdiv_r = dlsym(GMP.h, "__gmpz_fdiv_r");
    if (!GMP.init || !GMP.clear || !GMP.import || !GMP.export || !GMP.mul || !GMP.sub || !GMP.fdiv_r) { GMP.h = NULL; return 0; }
    return 1;
}
int orc_gmp_available(void) { return gmp_load(); }
/* out = (a * b mod x^n + 1) mod q; a, b, out raw u32[n][W] little-endian words, q as qW words.  -1 without GMP. */
int orc_gmp_mul_xn1(uint32_t *out, const uint32_t *a, const uint32_t *b, int n, int W, const uint32_t *qwords, int qW) {
    if (!gmp_load()) return -1;
    const int slot = (64 * W + ilog2(n) + 2 + 63) / 64;               /* 64-bit limbs per coefficient slot: no overlap of products */
    const size_t limbs = (size_t)n * slot;
    uint64_t *pa = (uint64_t *)calloc(limbs, 8), *pb = (uint64_t *)calloc(limbs, 8), *pc = (uint64_t *)calloc(2 * limbs + 2, 8);
    for (int i = 0; i < n; i++) { memcpy(pa + (size_t)i * slot, a + (size_t)i * W, 4 * (size_t)W); memcpy(pb + (size_t)i * slot, b + (size_t)i * W, 4 * (size_t)W); }
    gmp_mpz A, B, Cc, Q, x, y;
    GMP.init(A); GMP.init(B); GMP.init(Cc); GMP.init(Q); GMP.init(x); GMP.init(y);
    GMP.import(A, limbs, -1, 8, 0, 0, pa); GMP.import(B, limbs, -1, 8, 0, 0, pb);
    GMP.import(Q, (size_t)qW, -1, 4, 0, 0, qwords);
    GMP.mul(Cc, A, B);                                                /* the one big multiplication */
    size_t got = 0;
    GMP.export(pc, &got, -1, 8, 0, 0, Cc);
    uint32_t *w = (uint32_t *)calloc((size_t)W + 2, 4);
    for (int i = 0; i < n; i++) {                                     /* c[i] - c[i + n] mod q   (x^n = -1) */
        GMP.import(x, (size_t)slot, -1, 8, 0, 0, pc + (size_t)i * slot);
        GMP.import(y, (size_t)slot, -1, 8, 0, 0, pc + (size_t)(i + n) * slot);
        GMP.sub(x, x, y);
        GMP.fdiv_r(x, x, Q);
        size_t cnt = 0;
        memset(w, 0, 4 * ((size_t)W + 2));
        GMP.export(w, &cnt, -1, 4, 0, 0, x);
        memcpy(out + (size_t)i * W, w, 4 * (size_t)W);
    }
    free(w); free(pa); free(pb); free(pc);
    GMP.clear(A); GMP.clear(B); GMP.clear(Cc); GMP.clear(Q); GMP.clear(x); GMP.clear(y);
    return 0;
}

uint64_t orc_splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_fill_u32_below(uint32_t *dst, size_t n, uint32_t bound, uint64_t seed) {
    uint64_t st = seed;
    for (size_t i = 0; i < n; i++) dst[i] = (uint32_t)(orc_splitmix64(&st) % bound);
}
