// mini_ntl/NTL/ZZ.h -- FALLBACK ONLY.  A tiny subset of NTL's ZZ (signed big
// integer) so that this repository's C++ API layer (cuhe_amd/cxx/CuHE.h) can be
// compiled and tested on machines without NTL.  When the real NTL is installed
// its headers are found first (see cuhe_amd/cxx/Makefile) and this directory is
// not on the include path.  Only what CuHE.h / the tests need is provided.
#pragma once
#include <sys/random.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#define NTL_CLIENT using namespace std; using namespace NTL;
#define CUHE_MINI_NTL 1

namespace NTL {

class ZZ {
public:
    std::vector<uint32_t> m;   // magnitude, little endian, no leading zeros
    bool neg = false;
    ZZ() {}
    ZZ(long v) { set(v); }
    void set(long v) {
        m.clear(); neg = v < 0;
        unsigned long u = neg ? 0UL - (unsigned long)v : (unsigned long)v;
        while (u) { m.push_back((uint32_t)u); u >>= 32; }
    }
    bool zero() const { return m.empty(); }
    void trim() { while (!m.empty() && m.back() == 0) m.pop_back(); if (m.empty()) neg = false; }
    static int cmpmag(const ZZ &a, const ZZ &b) {
        if (a.m.size() != b.m.size()) return a.m.size() < b.m.size() ? -1 : 1;
        for (size_t i = a.m.size(); i-- > 0;) if (a.m[i] != b.m[i]) return a.m[i] < b.m[i] ? -1 : 1;
        return 0;
    }
    static ZZ addmag(const ZZ &a, const ZZ &b) {
        ZZ r; uint64_t c = 0; size_t n = std::max(a.m.size(), b.m.size());
        for (size_t i = 0; i < n; ++i) { c += (i < a.m.size() ? a.m[i] : 0); c += (i < b.m.size() ? b.m[i] : 0); r.m.push_back((uint32_t)c); c >>= 32; }
        if (c) r.m.push_back((uint32_t)c);
        return r;
    }
    static ZZ submag(const ZZ &a, const ZZ &b) {   // |a| >= |b|
        ZZ r; int64_t br = 0;
        for (size_t i = 0; i < a.m.size(); ++i) {
            int64_t t = (int64_t)a.m[i] - (i < b.m.size() ? b.m[i] : 0) - br;
            br = t < 0; if (t < 0) t += (1LL << 32);
            r.m.push_back((uint32_t)t);
        }
        r.trim(); return r;
    }
    static ZZ add(const ZZ &a, const ZZ &b) {
        if (a.neg == b.neg) { ZZ r = addmag(a, b); r.neg = a.neg; r.trim(); return r; }
        int c = cmpmag(a, b);
        if (c == 0) return ZZ();
        ZZ r = c > 0 ? submag(a, b) : submag(b, a);
        r.neg = c > 0 ? a.neg : b.neg; r.trim(); return r;
    }
    static ZZ mul(const ZZ &a, const ZZ &b) {
        ZZ r; if (a.zero() || b.zero()) return r;
        r.m.assign(a.m.size() + b.m.size(), 0);
        for (size_t i = 0; i < a.m.size(); ++i) {
            uint64_t c = 0;
            for (size_t j = 0; j < b.m.size(); ++j) { uint64_t t = (uint64_t)a.m[i] * b.m[j] + r.m[i + j] + c; r.m[i + j] = (uint32_t)t; c = t >> 32; }
            r.m[i + b.m.size()] += (uint32_t)c;
        }
        r.neg = a.neg != b.neg; r.trim(); return r;
    }
    int bits() const { if (m.empty()) return 0; int n = 0; uint32_t t = m.back(); while (t) { ++n; t >>= 1; } return (int)(m.size() - 1) * 32 + n; }
    bool bit(int i) const { size_t w = (size_t)i / 32; return w < m.size() && ((m[w] >> (i % 32)) & 1); }
    void shl1() { uint32_t c = 0; for (auto &x : m) { uint32_t n = x >> 31; x = (x << 1) | c; c = n; } if (c) m.push_back(c); }
    // floor division (NTL semantics: remainder has the sign of the divisor)
    // |a| / |b| by schoolbook long division in base 2^32 (Knuth's algorithm D); b != 0
    static void divmag(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b, std::vector<uint32_t> &q, std::vector<uint32_t> &r) {
        const size_t n = b.size();
        if (a.size() < n) { q.clear(); r = a; return; }
        if (n == 1) {
            uint64_t rem = 0; q.assign(a.size(), 0);
            for (size_t i = a.size(); i-- > 0;) { const uint64_t cur = (rem << 32) | a[i]; q[i] = (uint32_t)(cur / b[0]); rem = cur % b[0]; }
            r.assign(1, (uint32_t)rem);
            return;
        }
        const size_t m = a.size() - n;
        int s = 0; for (uint32_t t = b.back(); !(t & 0x80000000u); t <<= 1) ++s;
        std::vector<uint32_t> v(n), u(a.size() + 1);
        for (size_t i = n; i-- > 1;) v[i] = (b[i] << s) | (s ? b[i - 1] >> (32 - s) : 0);
        v[0] = b[0] << s;
        u[a.size()] = s ? a.back() >> (32 - s) : 0;
        for (size_t i = a.size(); i-- > 1;) u[i] = (a[i] << s) | (s ? a[i - 1] >> (32 - s) : 0);
        u[0] = a[0] << s;
        q.assign(m + 1, 0);
        for (size_t j = m + 1; j-- > 0;) {
            const uint64_t num = ((uint64_t)u[j + n] << 32) | u[j + n - 1];
            uint64_t qhat = num / v[n - 1], rhat = num % v[n - 1];
            while (qhat >> 32 || qhat * v[n - 2] > ((rhat << 32) | u[j + n - 2])) { --qhat; rhat += v[n - 1]; if (rhat >> 32) break; }
            int64_t borrow = 0; uint64_t carry = 0;
            for (size_t i = 0; i < n; ++i) {
                const uint64_t pr = qhat * v[i] + carry; carry = pr >> 32;
                const int64_t t = (int64_t)u[i + j] - borrow - (int64_t)(pr & 0xffffffffu);
                u[i + j] = (uint32_t)t; borrow = t < 0;
            }
            const int64_t t = (int64_t)u[j + n] - borrow - (int64_t)carry;
            u[j + n] = (uint32_t)t;
            if (t < 0) {                       // qhat was one too large: add the divisor back
                --qhat; uint64_t c = 0;
                for (size_t i = 0; i < n; ++i) { c += (uint64_t)u[i + j] + v[i]; u[i + j] = (uint32_t)c; c >>= 32; }
                u[j + n] += (uint32_t)c;
            }
            q[j] = (uint32_t)qhat;
        }
        r.assign(n, 0);
        for (size_t i = 0; i < n; ++i) r[i] = (u[i] >> s) | (s ? u[i + 1] << (32 - s) : 0);
    }
    // floor division (NTL semantics: remainder has the sign of the divisor)
    static void divrem(const ZZ &a, const ZZ &b, ZZ &q, ZZ &r) {
        ZZ B = b; B.neg = false;
        q = ZZ(); r = ZZ();
        divmag(a.m, B.m, q.m, r.m);
        q.trim(); r.trim();
        if (a.neg != b.neg) {         // truncated -> floor
            q.neg = !q.zero();
            if (!r.zero()) { q = add(q, ZZ(-1)); r = submag(B, r); }
        }
        if (!r.zero()) r.neg = b.neg;
    }
};

inline ZZ operator+(const ZZ &a, const ZZ &b) { return ZZ::add(a, b); }
inline ZZ operator-(const ZZ &a) { ZZ r = a; if (!r.zero()) r.neg = !r.neg; return r; }
inline ZZ operator-(const ZZ &a, const ZZ &b) { return ZZ::add(a, -b); }
inline ZZ operator*(const ZZ &a, const ZZ &b) { return ZZ::mul(a, b); }
inline ZZ operator/(const ZZ &a, const ZZ &b) { ZZ q, r; ZZ::divrem(a, b, q, r); return q; }
inline ZZ operator%(const ZZ &a, const ZZ &b) { ZZ q, r; ZZ::divrem(a, b, q, r); return r; }
inline ZZ &operator+=(ZZ &a, const ZZ &b) { a = a + b; return a; }
inline ZZ &operator-=(ZZ &a, const ZZ &b) { a = a - b; return a; }
inline ZZ &operator*=(ZZ &a, const ZZ &b) { a = a * b; return a; }
inline ZZ &operator%=(ZZ &a, const ZZ &b) { a = a % b; return a; }
inline ZZ &operator/=(ZZ &a, const ZZ &b) { a = a / b; return a; }
inline int compare(const ZZ &a, const ZZ &b) {
    if (a.neg != b.neg) return a.neg ? -1 : 1;
    int c = ZZ::cmpmag(a, b); return a.neg ? -c : c;
}
inline bool operator==(const ZZ &a, const ZZ &b) { return compare(a, b) == 0; }
inline bool operator!=(const ZZ &a, const ZZ &b) { return compare(a, b) != 0; }
inline bool operator<(const ZZ &a, const ZZ &b) { return compare(a, b) < 0; }
inline bool operator>(const ZZ &a, const ZZ &b) { return compare(a, b) > 0; }
inline bool operator<=(const ZZ &a, const ZZ &b) { return compare(a, b) <= 0; }
inline bool operator>=(const ZZ &a, const ZZ &b) { return compare(a, b) >= 0; }

inline ZZ to_ZZ(long v) { return ZZ(v); }
inline ZZ to_ZZ(int v) { return ZZ((long)v); }
inline ZZ to_ZZ(unsigned v) { return ZZ((long)v); }
inline ZZ to_ZZ(unsigned long v) { ZZ r; while (v) { r.m.push_back((uint32_t)v); v >>= 32; } return r; }
inline ZZ to_ZZ(const char *s) {
    ZZ r; bool neg = false; if (*s == '-') { neg = true; ++s; }
    for (; *s; ++s) r = r * ZZ(10) + ZZ(*s - '0');
    if (neg) r = -r;
    return r;
}
inline long to_long(const ZZ &a) { unsigned long v = 0; for (size_t i = a.m.size(); i-- > 0;) v = (v << 32) | a.m[i]; return a.neg ? -(long)v : (long)v; }
inline void conv(unsigned &x, const ZZ &a) { x = a.m.empty() ? 0u : a.m[0]; }
inline void conv(long &x, const ZZ &a) { x = to_long(a); }
inline void conv(ZZ &x, long a) { x = ZZ(a); }
inline void conv(ZZ &x, int a) { x = ZZ((long)a); }
inline void conv(ZZ &x, const char *s) { x = to_ZZ(s); }
template <class T, class S> inline T conv(const S &a) { T x; conv(x, a); return x; }
inline long NumBits(const ZZ &a) { return a.bits(); }
inline long IsZero(const ZZ &a) { return a.zero(); }
inline long IsOdd(const ZZ &a) { return !a.m.empty() && (a.m[0] & 1); }
inline long bit(const ZZ &a, long k) { return a.bit((int)k); }
inline void clear(ZZ &a) { a = ZZ(); }
inline ZZ power(const ZZ &a, long e) { ZZ r(1), b = a; while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; } return r; }
inline ZZ operator<<(const ZZ &a, long k) {
    if (a.zero() || k == 0) return a;
    ZZ r; r.neg = a.neg; r.m.assign(a.m.size() + (size_t)k / 32 + 1, 0);
    const int s = (int)(k % 32); const size_t w = (size_t)k / 32;
    for (size_t i = 0; i < a.m.size(); ++i) { const uint64_t v = (uint64_t)a.m[i] << s; r.m[i + w] |= (uint32_t)v; r.m[i + w + 1] |= (uint32_t)(v >> 32); }
    r.trim(); return r;
}
inline ZZ operator>>(const ZZ &a, long k) {          // magnitude shift, sign kept (NTL semantics)
    ZZ r; const size_t w = (size_t)k / 32; const int s = (int)(k % 32);
    if (w >= a.m.size()) return r;
    r.neg = a.neg; r.m.assign(a.m.size() - w, 0);
    for (size_t i = w; i < a.m.size(); ++i) { uint64_t v = a.m[i]; if (i + 1 < a.m.size()) v |= (uint64_t)a.m[i + 1] << 32; r.m[i - w] = (uint32_t)(v >> s); }
    r.trim(); return r;
}
inline long GCD(long a, long b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { const long t = a % b; a = b; b = t; } return a; }
inline ZZ GCD(const ZZ &a, const ZZ &b) { ZZ x = a, y = b; x.neg = y.neg = false; while (!y.zero()) { ZZ t = x % y; x = y; y = t; } return x; }
// deterministic Miller-Rabin for 64-bit values (the bases 2..37 decide every n < 2^64)
inline long ProbPrime(long n, long = 10) {
    if (n < 2) return 0;
    for (long p : {2L, 3L, 5L, 7L, 11L, 13L, 17L, 19L, 23L, 29L, 31L, 37L}) { if (n == p) return 1; if (n % p == 0) return 0; }
    typedef unsigned __int128 u128; const uint64_t N = (uint64_t)n; uint64_t d = N - 1; int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (uint64_t a : {2ULL, 3ULL, 5ULL, 7ULL, 11ULL, 13ULL, 17ULL, 19ULL, 23ULL, 29ULL, 31ULL, 37ULL}) {
        uint64_t x = 1, b = a % N, e = d;
        while (e) { if (e & 1) x = (uint64_t)((u128)x * b % N); b = (uint64_t)((u128)b * b % N); e >>= 1; }
        if (x == 1 || x == N - 1) continue;
        bool comp = true;
        for (int i = 1; i < r && comp; ++i) { x = (uint64_t)((u128)x * x % N); if (x == N - 1) comp = false; }
        if (comp) return 0;
    }
    return 1;
}
inline ZZ power2_ZZ(long e) { ZZ r; r.m.assign(e / 32 + 1, 0); r.m[e / 32] = 1u << (e % 32); return r; }
// BytesFromZZ: little-endian bytes of |a|, zero padded / truncated to n (NTL semantics).  The magnitude is an
// array of little-endian 32-bit words, so on a little-endian host this is a memcpy.
inline void BytesFromZZ(unsigned char *p, const ZZ &a, long n) {
    const long have = std::min<long>(n, (long)a.m.size() * 4);
    if (have) std::memcpy(p, a.m.data(), (size_t)have);
    if (n > have) std::memset(p + have, 0, (size_t)(n - have));
}
inline void ZZFromBytes(ZZ &r, const unsigned char *p, long n) {
    r.neg = false; r.m.assign((size_t)(n + 3) / 4, 0);
    if (n) std::memcpy(r.m.data(), p, (size_t)n);
    r.trim();
}
inline ZZ ZZFromBytes(const unsigned char *p, long n) { ZZ r; ZZFromBytes(r, p, n); return r; }
inline std::ostream &operator<<(std::ostream &os, const ZZ &a) {
    if (a.zero()) return os << "0";
    std::string s; ZZ t = a; t.neg = false; ZZ ten(10);
    while (!t.zero()) { ZZ q, r; ZZ::divrem(t, ten, q, r); s.push_back((char)('0' + (r.m.empty() ? 0 : r.m[0]))); t = q; }
    if (a.neg) s.push_back('-');
    std::reverse(s.begin(), s.end());
    return os << s;
}
// Random source of this fallback.  NOT a cryptographic PRG: until SetSeed is called every word comes from the
// operating system (getrandom), so a client built against mini_ntl by accident does not draw predictable keys; after
// SetSeed(s) -- which the tests call for reproducible runs -- it is a 64-bit xorshift stream determined by s alone.
// Deployments use the real NTL (Makefile: NTL_PREFIX); INTEGRATION.md says so.
inline uint64_t &mini_seed() { static uint64_t s = 0; return s; }
inline bool &mini_seeded() { static bool b = false; return b; }
inline void SetSeed(const ZZ &s) { mini_seed() = (uint64_t)to_long(s) * 2654435761ULL + 1; mini_seeded() = true; }
inline uint64_t mini_next() {
    if (!mini_seeded()) {
        uint64_t v = 0; size_t got = 0;
        while (got < sizeof v) {
            const ssize_t r = getrandom((unsigned char *)&v + got, sizeof v - got, 0);
            if (r <= 0) { std::fprintf(stderr, "mini_ntl: getrandom failed\n"); std::abort(); }
            got += (size_t)r;
        }
        return v;
    }
    uint64_t &s = mini_seed(); s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s;
}
inline ZZ RandomBnd(const ZZ &n) {
    ZZ r; r.m.assign(n.m.size() + 1, 0);
    for (auto &x : r.m) x = (uint32_t)mini_next();
    r.trim(); return r % n;
}
inline ZZ RandomBits_ZZ(long l) { return RandomBnd(power2_ZZ(l)); }

}  // namespace NTL
