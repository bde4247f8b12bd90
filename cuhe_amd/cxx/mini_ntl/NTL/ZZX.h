// mini_ntl/NTL/ZZX.h -- FALLBACK ONLY (see ZZ.h): polynomial over ZZ, low-to-high.
#pragma once
#include "ZZ.h"

namespace NTL {

class ZZX {
public:
    std::vector<ZZ> rep;
    ZZX() {}
    ZZX(const ZZX &o) : rep(o.rep) {}
    ZZX &operator=(const ZZX &o) { rep = o.rep; return *this; }
    ZZX(ZZX &&o) noexcept : rep(std::move(o.rep)) {}
    ZZX &operator=(ZZX &&o) noexcept { rep = std::move(o.rep); return *this; }
    ZZX &operator=(long c) { rep.clear(); if (c) rep.push_back(ZZ(c)); return *this; }      // constant polynomial
    // releases the buffer and leaves a valid empty vector behind, so that the explicit-destructor-then-scope-exit
    // pattern of the reference's examples (Prince.cu:298-319) stays harmless with this fallback type too
    ~ZZX() { std::vector<ZZ>().swap(rep); }
    void normalize() { while (!rep.empty() && rep.back().zero()) rep.pop_back(); }
};
inline long deg(const ZZX &a) { return (long)a.rep.size() - 1; }
inline const ZZ &coeff(const ZZX &a, long i) { static const ZZ z; return (i < 0 || i >= (long)a.rep.size()) ? z : a.rep[i]; }
inline void SetCoeff(ZZX &a, long i, const ZZ &v) { if (i >= (long)a.rep.size()) a.rep.resize(i + 1); a.rep[i] = v; a.normalize(); }
inline void SetCoeff(ZZX &a, long i, long v) { SetCoeff(a, i, ZZ(v)); }
inline void SetCoeff(ZZX &a, long i) { SetCoeff(a, i, ZZ(1)); }
inline void clear(ZZX &a) { std::vector<ZZ>().swap(a.rep); }
inline bool operator==(const ZZX &a, const ZZX &b) {
    if (a.rep.size() != b.rep.size()) return false;
    for (size_t i = 0; i < a.rep.size(); ++i) if (a.rep[i] != b.rep[i]) return false;
    return true;
}
inline bool operator!=(const ZZX &a, const ZZX &b) { return !(a == b); }
inline ZZX operator+(const ZZX &a, const ZZX &b) {
    ZZX r; r.rep.resize(std::max(a.rep.size(), b.rep.size()));
    for (size_t i = 0; i < r.rep.size(); ++i) r.rep[i] = coeff(a, i) + coeff(b, i);
    r.normalize(); return r;
}
inline ZZX operator-(const ZZX &a, const ZZX &b) {
    ZZX r; r.rep.resize(std::max(a.rep.size(), b.rep.size()));
    for (size_t i = 0; i < r.rep.size(); ++i) r.rep[i] = coeff(a, i) - coeff(b, i);
    r.normalize(); return r;
}
inline ZZX operator*(const ZZX &a, const ZZX &b) {       // schoolbook: tests only
    ZZX r; if (a.rep.empty() || b.rep.empty()) return r;
    r.rep.assign(a.rep.size() + b.rep.size() - 1, ZZ());
    for (size_t i = 0; i < a.rep.size(); ++i) if (!a.rep[i].zero())
        for (size_t j = 0; j < b.rep.size(); ++j) if (!b.rep[j].zero()) r.rep[i + j] += a.rep[i] * b.rep[j];
    r.normalize(); return r;
}
// remainder modulo a MONIC polynomial
inline ZZX operator%(const ZZX &a, const ZZX &m) {
    ZZX r = a; long n = deg(m);
    for (long k = deg(r); k >= n; --k) {
        ZZ c = coeff(r, k);
        if (c.zero()) continue;
        for (long i = 0; i <= n; ++i) if (!m.rep[i].zero()) r.rep[k - n + i] -= c * m.rep[i];
    }
    r.normalize(); return r;
}
inline ZZX &operator%=(ZZX &a, const ZZX &m) { a = a % m; return a; }
inline ZZX operator*(const ZZX &a, const ZZ &c) { ZZX r; r.rep.resize(a.rep.size()); for (size_t i = 0; i < a.rep.size(); ++i) r.rep[i] = a.rep[i] * c; r.normalize(); return r; }
inline ZZX operator*(const ZZ &c, const ZZX &a) { return a * c; }
inline ZZX operator*(const ZZX &a, long c) { return a * ZZ(c); }
inline ZZX operator*(long c, const ZZX &a) { return a * ZZ(c); }
inline ZZX &operator*=(ZZX &a, const ZZ &c) { a = a * c; return a; }
inline ZZX &operator*=(ZZX &a, long c) { a = a * ZZ(c); return a; }
inline ZZX operator+(const ZZX &a, long c) { ZZX r = a; if (r.rep.empty()) r.rep.resize(1); r.rep[0] += ZZ(c); r.normalize(); return r; }
inline ZZX operator-(const ZZX &a) { ZZX r = a; for (auto &c : r.rep) c = -c; return r; }
inline ZZX &operator-=(ZZX &a, const ZZX &b) { a = a - b; return a; }
// quotient by a polynomial whose leading coefficient is +-1 (what the cyclotomic construction of the examples needs)
inline ZZX operator/(const ZZX &a, const ZZX &b) {
    ZZX q, r = a; const long n = deg(b);
    if (n < 0 || deg(a) < n) return q;
    const bool negLead = b.rep[n] == ZZ(-1);
    q.rep.assign((size_t)(deg(a) - n) + 1, ZZ());
    for (long k = deg(r); k >= n; --k) {
        ZZ c = coeff(r, k); if (c.zero()) continue;
        if (negLead) c = -c;
        q.rep[k - n] = c;
        for (long i = 0; i <= n; ++i) if (!b.rep[i].zero()) r.rep[k - n + i] -= c * b.rep[i];
    }
    q.normalize(); return q;
}
inline ZZX &operator/=(ZZX &a, const ZZX &b) { a = a / b; return a; }
inline ZZX &operator+=(ZZX &a, const ZZX &b) { a = a + b; return a; }
inline ZZX &operator*=(ZZX &a, const ZZX &b) { a = a * b; return a; }
inline std::ostream &operator<<(std::ostream &os, const ZZX &a) {
    os << "["; for (size_t i = 0; i < a.rep.size(); ++i) os << (i ? " " : "") << a.rep[i]; return os << "]";
}

}  // namespace NTL
