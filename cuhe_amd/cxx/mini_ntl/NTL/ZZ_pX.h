// mini_ntl/NTL/ZZ_pX.h -- FALLBACK ONLY (see ZZ.h): polynomials over Z_p for the current ZZ_p modulus, the subset
// of NTL's ZZ_pX the DHS scheme client uses (examples/DHS/DHS.cu: key inverse, batching over GF(2)).
// Two storage forms: modulus 2 -> one bit per coefficient (the batching code divides a degree-16384 polynomial by
// 65536 candidate factors); any other modulus -> a vector of residues.
#pragma once
#include "ZZX.h"
#include "ZZ_p.h"

namespace NTL {

class ZZ_pX {
public:
    bool bin;                         // coefficients mod 2, bit packed in b
    std::vector<uint64_t> b;          // bit i = coefficient of x^i
    std::vector<ZZ> c;                // general modulus: residues in [0, p), no trailing zeros
    ZZ_pX() : bin(ZZ_p::modulus() == ZZ(2)) {}
    ZZ_pX &operator=(long a);
    void normalize() {
        if (bin) while (!b.empty() && b.back() == 0) b.pop_back();
        else while (!c.empty() && c.back().zero()) c.pop_back();
    }
};

// ---- bit-packed helpers
inline long bits_deg(const std::vector<uint64_t> &b) {
    for (size_t i = b.size(); i-- > 0;) if (b[i]) return (long)i * 64 + 63 - __builtin_clzll(b[i]);
    return -1;
}
inline bool bits_get(const std::vector<uint64_t> &b, long i) { return i >= 0 && (size_t)(i >> 6) < b.size() && ((b[i >> 6] >> (i & 63)) & 1); }
inline void bits_flip(std::vector<uint64_t> &b, long i) { if ((size_t)(i >> 6) >= b.size()) b.resize((i >> 6) + 1, 0); b[i >> 6] ^= 1ULL << (i & 63); }
// r ^= s << sh, s of degree sdeg
inline void bits_xor_shifted(std::vector<uint64_t> &r, const std::vector<uint64_t> &s, long sdeg, long sh) {
    if (sdeg < 0) return;
    const size_t need = (size_t)((sdeg + sh) >> 6) + 1;
    if (r.size() < need) r.resize(need, 0);
    const size_t w = (size_t)sh >> 6; const int k = (int)(sh & 63);
    const size_t sw = (size_t)(sdeg >> 6) + 1;
    for (size_t i = 0; i < sw; ++i) {
        r[i + w] ^= s[i] << k;
        if (k && (s[i] >> (64 - k))) r[i + w + 1] ^= s[i] >> (64 - k);
    }
}

inline long deg(const ZZ_pX &a) { return a.bin ? bits_deg(a.b) : (long)a.c.size() - 1; }
inline void clear(ZZ_pX &a) { a.b.clear(); a.c.clear(); }
inline long IsZero(const ZZ_pX &a) { return deg(a) < 0; }
inline ZZ_p coeff(const ZZ_pX &a, long i) {
    ZZ_p r;
    if (a.bin) { if (bits_get(a.b, i)) r.v = ZZ(1); }
    else if (i >= 0 && i < (long)a.c.size()) r.v = a.c[i];
    return r;
}
inline void SetCoeff(ZZ_pX &x, long i, const ZZ &a) {
    if (x.bin) { if (bits_get(x.b, i) != (bool)IsOdd(a)) bits_flip(x.b, i); }
    else { if (i >= (long)x.c.size()) x.c.resize(i + 1); x.c[i] = a % ZZ_p::modulus(); }
    x.normalize();
}
inline void SetCoeff(ZZ_pX &x, long i, const ZZ_p &a) { SetCoeff(x, i, a.v); }
inline void SetCoeff(ZZ_pX &x, long i, long a) { SetCoeff(x, i, ZZ(a)); }
inline void SetCoeff(ZZ_pX &x, long i) { SetCoeff(x, i, ZZ(1)); }
inline ZZ_pX &ZZ_pX::operator=(long a) { b.clear(); c.clear(); bin = ZZ_p::modulus() == ZZ(2); SetCoeff(*this, 0, a); return *this; }

inline ZZ_pX to_ZZ_pX(const ZZX &a) {
    ZZ_pX r;
    if (r.bin) { for (long i = 0; i <= deg(a); ++i) if (IsOdd(a.rep[i])) bits_flip(r.b, i); }
    else { r.c.resize(a.rep.size()); for (size_t i = 0; i < a.rep.size(); ++i) r.c[i] = a.rep[i] % ZZ_p::modulus(); }
    r.normalize(); return r;
}
inline void conv(ZZ_pX &x, const ZZX &a) { x = to_ZZ_pX(a); }
inline ZZX to_ZZX(const ZZ_pX &a) {
    ZZX r; const long d = deg(a);
    r.rep.resize(d + 1);
    for (long i = 0; i <= d; ++i) r.rep[i] = a.bin ? ZZ(bits_get(a.b, i) ? 1 : 0) : a.c[i];
    r.normalize(); return r;
}
inline void conv(ZZX &x, const ZZ_pX &a) { x = to_ZZX(a); }

inline bool operator==(const ZZ_pX &a, const ZZ_pX &b) {
    const long da = deg(a);
    if (da != deg(b)) return false;
    for (long i = 0; i <= da; ++i) if (!(coeff(a, i).v == coeff(b, i).v)) return false;
    return true;
}
inline bool operator!=(const ZZ_pX &a, const ZZ_pX &b) { return !(a == b); }

inline ZZ_pX operator+(const ZZ_pX &a, const ZZ_pX &b) {
    ZZ_pX r = a;
    if (a.bin) { bits_xor_shifted(r.b, b.b, bits_deg(b.b), 0); }
    else {
        if (r.c.size() < b.c.size()) r.c.resize(b.c.size());
        for (size_t i = 0; i < b.c.size(); ++i) { r.c[i] += b.c[i]; if (r.c[i] >= ZZ_p::modulus()) r.c[i] -= ZZ_p::modulus(); }
    }
    r.normalize(); return r;
}
inline ZZ_pX operator-(const ZZ_pX &a, const ZZ_pX &b) {
    if (a.bin) return a + b;
    ZZ_pX r = a;
    if (r.c.size() < b.c.size()) r.c.resize(b.c.size());
    for (size_t i = 0; i < b.c.size(); ++i) { r.c[i] -= b.c[i]; if (r.c[i] < ZZ(0)) r.c[i] += ZZ_p::modulus(); }
    r.normalize(); return r;
}
inline ZZ_pX operator*(const ZZ_pX &a, const ZZ_pX &b) {
    ZZ_pX r; r.bin = a.bin;
    const long da = deg(a), db = deg(b);
    if (da < 0 || db < 0) return r;
    if (a.bin) {
        const ZZ_pX &s = da <= db ? a : b, &l = da <= db ? b : a;     // few shifts of the long operand
        const long ds = deg(s), dl = deg(l);
        for (long i = 0; i <= ds; ++i) if (bits_get(s.b, i)) bits_xor_shifted(r.b, l.b, dl, i);
    } else {
        r.c.assign((size_t)da + db + 1, ZZ());
        for (long i = 0; i <= da; ++i) if (!a.c[i].zero()) for (long j = 0; j <= db; ++j) if (!b.c[j].zero()) r.c[i + j] += a.c[i] * b.c[j];
        for (auto &x : r.c) x %= ZZ_p::modulus();
    }
    r.normalize(); return r;
}
// a = q b + r, deg r < deg b; throws std::runtime_error if the leading coefficient of b is not invertible
inline void DivRem(ZZ_pX &q, ZZ_pX &r, const ZZ_pX &a, const ZZ_pX &b) {
    const long db = deg(b);
    if (db < 0) throw std::runtime_error("DivRem: division by zero");
    ZZ_pX quo, rem = a; quo.bin = rem.bin = a.bin;
    if (a.bin) {
        for (long i = bits_deg(rem.b); i >= db; --i)
            if (bits_get(rem.b, i)) { bits_flip(quo.b, i - db); bits_xor_shifted(rem.b, b.b, db, i - db); }
    } else {
        const ZZ &p = ZZ_p::modulus();
        const ZZ linv = b.c[db] == ZZ(1) ? ZZ(1) : InvMod(b.c[db], p);
        if (deg(a) >= db) quo.c.assign((size_t)(deg(a) - db) + 1, ZZ());
        for (long k = deg(rem); k >= db; --k) {
            if (rem.c[k].zero()) continue;
            const ZZ cq = (rem.c[k] * linv) % p;
            quo.c[k - db] = cq;
            for (long i = 0; i <= db; ++i) if (!b.c[i].zero()) { rem.c[k - db + i] = (rem.c[k - db + i] - cq * b.c[i]) % p; }
        }
    }
    quo.normalize(); rem.normalize();
    q = quo; r = rem;
}
inline ZZ_pX operator/(const ZZ_pX &a, const ZZ_pX &b) { ZZ_pX q, r; DivRem(q, r, a, b); return q; }
inline ZZ_pX operator%(const ZZ_pX &a, const ZZ_pX &b) { ZZ_pX q, r; DivRem(q, r, a, b); return r; }
inline ZZ_pX &operator%=(ZZ_pX &a, const ZZ_pX &b) { a = a % b; return a; }
inline ZZ_pX &operator/=(ZZ_pX &a, const ZZ_pX &b) { a = a / b; return a; }
inline ZZ_pX &operator+=(ZZ_pX &a, const ZZ_pX &b) { a = a + b; return a; }
inline ZZ_pX &operator-=(ZZ_pX &a, const ZZ_pX &b) { a = a - b; return a; }
inline ZZ_pX &operator*=(ZZ_pX &a, const ZZ_pX &b) { a = a * b; return a; }

class vec_ZZ_pX {
public:
    std::vector<ZZ_pX> v;
    void SetLength(long n) { v.resize((size_t)n); }
    long length() const { return (long)v.size(); }
    ZZ_pX &operator[](long i) { return v[(size_t)i]; }
    const ZZ_pX &operator[](long i) const { return v[(size_t)i]; }
};

}  // namespace NTL
