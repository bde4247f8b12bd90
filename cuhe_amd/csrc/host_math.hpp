// host_math.hpp -- host-side number theory for the precomputation that
// initCuHE performs with NTL in the reference (cuhe/Parameters.cu:53-85,
// cuhe/Operations.cu:37-134,213-238, cuhe/Base.cu:58-70).  Self-contained
// (own multiword unsigned integers) so the library has no NTL/GMP dependency;
// values are identical to the reference's (checked against tests/golden/).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace cuhe { namespace host {

typedef unsigned __int128 u128;
static const uint64_t P = 0xffffffff00000001ULL;
static const uint64_t G = 15893793146607301539ULL;   // cuhe/Base.cu:65

inline uint64_t mulP(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
inline uint64_t powP(uint64_t x, uint64_t e) {
    uint64_t r = 1; x %= P;
    while (e) { if (e & 1) r = mulP(r, x); x = mulP(x, x); e >>= 1; }
    return r;
}

// psi with psi^2 = w_len = G^(65536/len): a primitive 2*len-th root of unity, the twist of the negacyclic transform of
// length len <= 65536.  For len < 65536 it is a power of G.  For len = 65536 it is a square root of G, a primitive
// 2^17-th root (2^32 divides P - 1): with z a generator of the 2-Sylow subgroup of Z_P^* (order
// 2^32) G = z^e for an e divisible by 2^16, found bit by bit (Pohlig-Hellman), and psi = z^(e/2).  Returns 0 on failure.
inline uint64_t root_2len(int len) {
    if (len < 65536) return powP(G, (uint64_t)(32768 / len));
    uint64_t z = 0;
    for (uint64_t base : {7ULL, 3ULL, 5ULL, 11ULL, 13ULL}) {
        z = powP(base, (P - 1) >> 32);
        if (powP(z, 1ULL << 31) != 1) break;          // order exactly 2^32
        z = 0;
    }
    if (!z) return 0;
    const uint64_t zinv = powP(z, P - 2);
    uint64_t e = 0;
    for (int i = 0; i < 32; ++i) {
        uint64_t t = mulP(G, powP(zinv, e));          // order divides 2^(32-i)
        t = powP(t, 1ULL << (31 - i));
        if (t != 1) e |= 1ULL << i;
    }
    if (e & 1) return 0;
    const uint64_t psi = powP(z, e >> 1);
    return mulP(psi, psi) == G ? psi : 0;
}

inline int numbits(uint64_t x) { int n = 0; while (x) { ++n; x >>= 1; } return n; }
inline uint64_t isqrt(uint64_t x) {
    uint64_t r = 0, bit = 1ULL << 62;
    while (bit > x) bit >>= 2;
    while (bit) { if (x >= r + bit) { x -= r + bit; r = (r >> 1) + bit; } else r >>= 1; bit >>= 2; }
    return r;
}
inline long totient(long x) {                     // cuhe/Parameters.cu:35-52 (returns x for x < 3)
    if (x < 3) return x;
    long res = x, n = x;
    for (long t = 2; t * t <= n; ++t)
        if (n % t == 0) { while (n % t == 0) n /= t; res = res / t * (t - 1); }
    if (n > 1) res = res / n * (n - 1);
    return res;
}
inline uint32_t powmod32(uint32_t b, uint32_t e, uint32_t m) {
    uint64_t r = 1, x = b % m;
    while (e) { if (e & 1) r = r * x % m; x = x * x % m; e >>= 1; }
    return (uint32_t)r;
}
inline bool is_prime32(uint32_t n) {              // exact below 2^32 (ProbPrime(.,10) in the reference)
    if (n < 2) return false;
    for (uint32_t q : {2u, 3u, 5u, 7u, 11u, 13u, 17u, 19u, 23u, 29u, 31u, 37u}) {
        if (n == q) return true;
        if (n % q == 0) return false;
    }
    uint32_t d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (uint32_t a : {2u, 7u, 61u}) {
        uint64_t x = powmod32(a % n, d, n);
        if (a % n == 0 || x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int r = 1; r < s; ++r) { x = x * x % n; if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}
inline uint32_t invmod32(uint32_t a, uint32_t m) {
    long long t = 0, nt = 1, r = m, nr = a % m;
    while (nr) { long long q = r / nr, x = t - q * nt; t = nt; nt = x; x = r - q * nr; r = nr; nr = x; }
    if (t < 0) t += m;
    return (uint32_t)t;
}

// little-endian multiword unsigned
struct BigU {
    std::vector<uint32_t> w;
    BigU() {}
    explicit BigU(uint32_t v) : w(1, v) {}
    void trim() { while (w.size() > 1 && w.back() == 0) w.pop_back(); }
    void mul_small(uint32_t m) {
        uint64_t c = 0;
        for (auto &x : w) { uint64_t t = (uint64_t)x * m + c; x = (uint32_t)t; c = t >> 32; }
        if (c) w.push_back((uint32_t)c);
    }
    BigU div_small(uint32_t d, uint32_t *rem = nullptr) const {
        BigU q; q.w.assign(w.size(), 0);
        uint64_t r = 0;
        for (int i = (int)w.size() - 1; i >= 0; --i) { uint64_t t = (r << 32) | w[i]; q.w[i] = (uint32_t)(t / d); r = t % d; }
        if (rem) *rem = (uint32_t)r;
        q.trim();
        return q;
    }
    uint32_t mod_small(uint32_t d) const {
        uint64_t r = 0;
        for (int i = (int)w.size() - 1; i >= 0; --i) r = ((r << 32) | w[i]) % d;
        return (uint32_t)r;
    }
    void to_words(uint32_t *dst, int n) const {
        for (int i = 0; i < n; ++i) dst[i] = i < (int)w.size() ? w[i] : 0;
    }
};

struct Params {                                   // cuhe/Parameters.h:34-62
    int mSize = 0, modLen = 0, modLen2 = 0, rawLen = 0, crtLen = 0, nttLen = 0;
    int logCoeffMax = 0, logCoeffMin = 0, logCoeffCut = 0;
    int depth = 0, modMsg = 0, logMsg = 0, wordsMsg = 0;
    int logRelin = 0, numEvalKey = 0;
    int logCrtPrime = 0, numCrtPrime = 0;

    void set(int d, int p, int w, int min, int cut, int m) {          // Parameters.cu:53-85
        depth = d; modMsg = p; logRelin = w; logCoeffMin = min; logCoeffCut = cut; mSize = m;
        logCoeffMax = min + cut * (d - 1);
        modLen = (int)totient(m);
        modLen2 = 1 << numbits((uint64_t)modLen - 1);
        if (modLen2 < 8192) modLen2 = 8192;
        rawLen = crtLen = modLen2;
        nttLen = 2 * modLen2;
        logMsg = numbits((uint64_t)p - 1);
        wordsMsg = (logMsg + 31) / 32;
        numEvalKey = w ? (logCoeffMax + w - 1) / w : 0;
        logCrtPrime = numbits(isqrt(P / (uint64_t)modLen));
        if (ncOnly()) {
            // ring degree 2^16 (m = 2^17): beyond the reference, whose transforms stop at 65536 points = degree 2^15
            // (cuhe/Parameters.cu:63-68, Base.cu:59-62).  Only the negacyclic representation exists: transforms of
            // modLen points, and the primes obey the centred-lift bound 2 n p^2 < P, i.e. at most 23 bits.
            nttLen = modLen2;
            logCrtPrime = numbits(isqrt(P / (2 * (uint64_t)modLen))) - 1;
        }
        numCrtPrime = (min + logCrtPrime - 1) / logCrtPrime;
        logCrtPrime = 0;
        while (logCrtPrime * numCrtPrime < min) ++logCrtPrime;
        numCrtPrime += d - 1;
    }
    bool ncOnly() const { return mSize == 131072; }
    int numCrtPrimeAt(int lvl) const { return lvl == -1 ? 1 : numCrtPrime - lvl; }        // :107-116
    int logCoeff(int lvl) const {                                                          // :117-128
        if (lvl == -1) return logMsg;
        if (lvl < depth) return logCoeffMax - lvl * logCoeffCut;
        return logCoeffMin - logCrtPrime;
    }
    int wordsCoeff(int lvl) const { int t = (logCoeff(lvl) + 31) / 32; return t > 1 ? t : 1; }  // :129-132
    int numEvalKeyAt(int lvl) const { return logRelin ? (logCoeff(lvl) + logRelin - 1) / logRelin : 0; } // :133-135
    int getLevel(int logq) const {                                                         // :136-141
        if (logq >= logCoeffMin) return (logCoeffMax - logq) / logCoeffCut;
        return -1;
    }
};

inline std::vector<uint32_t> gen_crt_primes(const Params &q) {         // cuhe/Operations.cu:37-80
    const int pnum = q.numCrtPrime, d = q.depth, l = q.logCrtPrime;
    std::vector<uint32_t> pr(pnum);
    const int logmid = q.logCoeffMin - (pnum - d) * l;
    uint32_t temp = (uint32_t)((1u << l) - 1);
    for (int i = 0; i <= pnum - d - 1; ++i) { while (!is_prime32(temp)) --temp; pr[i] = temp--; }
    uint32_t tmid = (logmid != l) ? (uint32_t)((1u << logmid) - 1) : temp;
    while (!is_prime32(tmid)) --tmid;
    pr[pnum - d] = tmid;
    if (q.logCoeffCut == logmid) temp = tmid - 1;
    else if (q.logCoeffCut == l) --temp;
    else temp = (uint32_t)((1u << q.logCoeffCut) - 1);
    for (int i = pnum - d + 1; i < pnum; ++i) {
        while (!is_prime32(temp) || temp % (uint32_t)q.modMsg != 1) --temp;
        pr[i] = temp--;
    }
    return pr;
}

inline int mobius(int n) {
    int mu = 1;
    for (int p = 2; p * p <= n; ++p)
        if (n % p == 0) { n /= p; if (n % p == 0) return 0; mu = -mu; }
    if (n > 1) mu = -mu;
    return mu;
}
// Phi_m, low-to-high coefficients (the DHS example builds this on the host with NTL)
inline std::vector<int32_t> cyclotomic(int m) {
    std::vector<long long> a(2 * (size_t)m + 2, 0);
    int len = 1; a[0] = 1;
    for (int d = 1; d <= m; ++d) {
        if (m % d || mobius(m / d) != 1) continue;
        for (int i = len - 1; i >= 0; --i) { a[i + d] += a[i]; a[i] = -a[i]; }
        len += d;
    }
    for (int d = 1; d <= m; ++d) {
        if (m % d || mobius(m / d) != -1) continue;
        for (int i = 0; i < len - d; ++i) a[i] = (i >= d ? a[i - d] : 0) - a[i];
        len -= d;
    }
    std::vector<int32_t> out(len);
    for (int i = 0; i < len; ++i) out[i] = (int32_t)a[i];
    return out;
}

// quotient of x^(2n-1) by the monic degree-n polynomial `mod` over the integers
// (cuhe/Operations.cu:217-219: SetCoeff(zu, 2n-1, 1); zu /= zm).  Returns false if a
// coefficient leaves the int64 safe range (never for cyclotomic moduli).
inline bool barrett_u(const std::vector<int32_t> &mod, std::vector<long long> &u) {
    const int n = (int)mod.size() - 1;
    std::vector<std::pair<int, long long>> nz;
    for (int i = 0; i < n; ++i) if (mod[i]) nz.push_back({i, mod[i]});
    std::vector<long long> rem(2 * (size_t)n, 0);
    rem[2 * n - 1] = 1;
    u.assign(n, 0);
    const long long LIM = 1LL << 40;
    for (int k = 2 * n - 1; k >= n; --k) {
        const long long c = rem[k];
        u[k - n] = c;
        if (!c) continue;
        if (c > LIM || c < -LIM) return false;
        for (auto &t : nz) rem[k - n + t.first] -= c * t.second;
    }
    return true;
}

// in-place length-`len` transform X[i] = sum_j a[j] w^(ij), w = G^(65536/len), natural order in and out (the
// device transform's definition, tests/test_ntt.cu:44-55), for tables built at init time
inline void ntt_host(std::vector<uint64_t> &a, int len) {
    int lg = 0; while ((1 << lg) < len) ++lg;
    for (int i = 0; i < len; ++i) {
        int r = 0; for (int b = 0; b < lg; ++b) if (i & (1 << b)) r |= 1 << (lg - 1 - b);
        if (i < r) std::swap(a[i], a[r]);
    }
    for (int half = 1; half < len; half <<= 1) {
        const uint64_t wl = powP(G, (uint64_t)(65536 / (2 * half)));
        for (int s0 = 0; s0 < len; s0 += 2 * half) {
            uint64_t w = 1;
            for (int j = 0; j < half; ++j) {
                const uint64_t u = a[s0 + j], v = mulP(a[s0 + j + half], w);
                uint64_t x = u + v; if (x < u || x >= P) x -= P;
                a[s0 + j] = x;
                a[s0 + j + half] = u >= v ? u - v : u + (P - v);
                w = mulP(w, wl);
            }
        }
    }
}

inline uint32_t smod(long long v, uint32_t p) { long long r = v % (long long)p; return (uint32_t)(r < 0 ? r + p : r); }

}}  // namespace cuhe::host
