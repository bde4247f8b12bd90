// mini_ntl/NTL/ZZ_pE.h -- FALLBACK ONLY (see ZZ.h): the ring Z_p[x]/(P) for the current ZZ_p modulus and a current
// polynomial modulus P -- the subset of NTL's ZZ_pE the DHS scheme client uses (to_ZZ_pE, inv, rep).
//
// inv() is the one heavy operation: the DHS key generation inverts a degree-(n-1) polynomial modulo the cyclotomic
// polynomial over Z_q, q a product of word-sized primes (examples/DHS/DHS.cu:372-387).  Z_q[x]/(P) is the product
// of the F_p[x]/(P) over the prime factors p of q, so when q is a squarefree product of primes below 2^31 (it is,
// for every cuHE parameter set) the inverse is computed per prime with machine arithmetic -- one thread per prime
// -- and lifted by the CRT; otherwise the extended Euclidean algorithm runs over Z_q itself and, like NTL's, gives
// up (std::runtime_error) when it meets a leading coefficient that is not a unit.
#pragma once
#include "ZZ_pX.h"
#include <mutex>
#include <thread>

namespace NTL {

class ZZ_pE {
public:
    ZZ_pX r;
    ZZ_pE() {}
    static ZZ_pX &mod() { static thread_local ZZ_pX m; return m; }
    static void init(const ZZ_pX &P) { mod() = P; }
    static const ZZ_pX &modulus() { return mod(); }
};
inline const ZZ_pX &rep(const ZZ_pE &a) { return a.r; }
inline ZZ_pE to_ZZ_pE(const ZZ_pX &a) { ZZ_pE e; e.r = deg(a) >= deg(ZZ_pE::modulus()) ? a % ZZ_pE::modulus() : a; return e; }
inline void conv(ZZ_pE &x, const ZZ_pX &a) { x = to_ZZ_pE(a); }
inline ZZ_pE operator+(const ZZ_pE &a, const ZZ_pE &b) { ZZ_pE e; e.r = a.r + b.r; return e; }
inline ZZ_pE operator-(const ZZ_pE &a, const ZZ_pE &b) { ZZ_pE e; e.r = a.r - b.r; return e; }
inline ZZ_pE operator*(const ZZ_pE &a, const ZZ_pE &b) { ZZ_pE e; e.r = (a.r * b.r) % ZZ_pE::modulus(); return e; }
inline bool operator==(const ZZ_pE &a, const ZZ_pE &b) { return a.r == b.r; }

namespace mini_detail {
typedef unsigned long long u64;
inline u64 powmod64(u64 b, u64 e, u64 p) { u64 r = 1; b %= p; while (e) { if (e & 1) r = (u64)((unsigned __int128)r * b % p); b = (u64)((unsigned __int128)b * b % p); e >>= 1; } return r; }
inline uint32_t mod_small(const ZZ &a, uint32_t p) { u64 r = 0; for (size_t i = a.m.size(); i-- > 0;) r = ((r << 32) | a.m[i]) % p; return (uint32_t)r; }
// inverse of f modulo (phi, p) over F_p, p < 2^31, by the extended Euclidean algorithm; false if gcd(f, phi) != 1
inline bool invert_mod_prime(std::vector<uint32_t> &inv, const std::vector<uint32_t> &f, const std::vector<uint32_t> &phi, uint32_t p) {
    typedef std::vector<u64> Poly;
    auto trim = [](Poly &a) { while (!a.empty() && a.back() == 0) a.pop_back(); };
    Poly r0(phi.begin(), phi.end()), r1(f.begin(), f.end()), t0, t1(1, 1);
    trim(r0); trim(r1);
    while (!r1.empty()) {
        const u64 lead = powmod64(r1.back(), p - 2, p);
        while (r0.size() >= r1.size()) {
            const size_t sh = r0.size() - r1.size();
            const u64 cq = r0.back() * lead % p, nc = cq ? p - cq : 0;
            u64 *a = r0.data() + sh; const u64 *b = r1.data();
            for (size_t i = 0, e = r1.size(); i < e; ++i) a[i] = (a[i] + nc * b[i]) % p;
            if (t0.size() < t1.size() + sh) t0.resize(t1.size() + sh, 0);
            a = t0.data() + sh; b = t1.data();
            for (size_t i = 0, e = t1.size(); i < e; ++i) a[i] = (a[i] + nc * b[i]) % p;
            trim(r0);
            if (r0.empty()) break;
        }
        std::swap(r0, r1); std::swap(t0, t1);
    }
    if (r0.size() != 1) return false;
    const u64 g = powmod64(r0[0], p - 2, p);
    Poly t = t0; const size_t n = phi.size() - 1;
    for (size_t k = t.size(); k-- > n;) {
        const u64 cq = t[k]; if (!cq) continue;
        const u64 nc = p - cq;
        for (size_t i = 0; i <= n; ++i) t[k - n + i] = (t[k - n + i] + nc * phi[i]) % p;
    }
    inv.assign(n, 0);
    for (size_t i = 0; i < n && i < t.size(); ++i) inv[i] = (uint32_t)(t[i] * g % p);
    return true;
}
// q as a squarefree product of primes below 2^26 (trial division against a sieve built once), or false
inline bool factor_small(std::vector<uint32_t> &primes, const ZZ &q) {
    const uint32_t LIM = 1u << 26;
    static std::vector<uint32_t> table;                      // primes below LIM
    static std::once_flag once;
    std::call_once(once, [&] {
        std::vector<uint64_t> comp(LIM / 128 + 1, 0);        // one bit per odd number: bit k <-> 2k+1
        auto get = [&](uint32_t k) { return (comp[k >> 6] >> (k & 63)) & 1; };
        for (uint32_t i = 3; (uint64_t)i * i < LIM; i += 2) if (!get(i / 2)) for (uint64_t j = (uint64_t)i * i; j < LIM; j += 2 * i) comp[(j / 2) >> 6] |= 1ULL << ((j / 2) & 63);
        table.push_back(2);
        for (uint32_t i = 3; i < LIM; i += 2) if (!get(i / 2)) table.push_back(i);
    });
    ZZ rest = q; rest.neg = false;
    primes.clear();
    for (size_t k = table.size(); k-- > 0 && !(rest == ZZ(1));) {   // cuHE's primes sit just below a power of two: start high
        const uint32_t p = table[k];
        if (mod_small(rest, p) != 0) continue;
        rest = rest / ZZ((long)p);
        if (mod_small(rest, p) == 0) return false;           // not squarefree
        primes.push_back(p);
    }
    return rest == ZZ(1);
}
// extended Euclid over Z_q itself (slow path)
inline ZZ_pX invert_generic(const ZZ_pX &f, const ZZ_pX &P) {
    ZZ_pX r0 = P, r1 = f, t0, t1; t0.bin = t1.bin = f.bin; SetCoeff(t1, 0, 1);
    while (deg(r1) >= 0) {
        ZZ_pX q, r; DivRem(q, r, r0, r1);                    // throws if the leading coefficient is not a unit
        ZZ_pX t2 = t0 - q * t1;
        r0 = r1; r1 = r; t0 = t1; t1 = t2;
    }
    if (deg(r0) != 0) throw std::runtime_error("inv: not invertible");
    ZZ_pX s; s.bin = f.bin; SetCoeff(s, 0, inv(coeff(r0, 0)).v);
    return (t0 * s) % P;
}
}  // namespace mini_detail

inline ZZ_pE inv(const ZZ_pE &a) {
    using namespace mini_detail;
    const ZZ_pX &P = ZZ_pE::modulus();
    ZZ_pE out;
    if (a.r.bin) { out.r = invert_generic(a.r, P); return out; }      // GF(2): bit-packed, small factors in practice
    const ZZ &q = ZZ_p::modulus();
    std::vector<uint32_t> primes;
    if (!factor_small(primes, q)) { out.r = invert_generic(a.r, P); return out; }
    const long n = deg(P);
    const size_t np = primes.size();
    std::vector<std::vector<uint32_t>> rows(np);
    std::vector<char> ok(np, 0);
    std::vector<std::thread> pool;
    for (size_t i = 0; i < np; ++i) pool.emplace_back([&, i] {
        const uint32_t p = primes[i];
        std::vector<uint32_t> fp((size_t)n, 0), php((size_t)n + 1);
        for (long k = 0; k <= deg(a.r) && k < n; ++k) fp[k] = mod_small(a.r.c[k], p);
        for (long k = 0; k <= n; ++k) php[k] = mod_small(P.c[k], p);
        ok[i] = invert_mod_prime(rows[i], fp, php, p);
    });
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < np; ++i) if (!ok[i]) throw std::runtime_error("inv: not invertible");
    std::vector<ZZ> lift(np);
    for (size_t i = 0; i < np; ++i) {
        const ZZ mi = q / ZZ((long)primes[i]);
        lift[i] = (mi * ZZ((long)powmod64(mod_small(mi, primes[i]), primes[i] - 2, primes[i]))) % q;
    }
    out.r.bin = false;
    out.r.c.assign((size_t)n, ZZ());
    for (long k = 0; k < n; ++k) {
        ZZ v;
        for (size_t i = 0; i < np; ++i) if (rows[i][k]) v += lift[i] * ZZ((long)rows[i][k]);
        out.r.c[k] = v % q;
    }
    out.r.normalize();
    return out;
}

}  // namespace NTL
