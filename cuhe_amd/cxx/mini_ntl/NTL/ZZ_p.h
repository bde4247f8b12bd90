// mini_ntl/NTL/ZZ_p.h -- FALLBACK ONLY (see ZZ.h): integers modulo the current modulus (NTL's ZZ_p subset).
#pragma once
#include "ZZ.h"
#include <stdexcept>

namespace NTL {

// x = a^-1 mod n if gcd(a, n) = 1 (returns 0), else x = gcd (returns 1) -- NTL's InvModStatus
inline long InvModStatus(ZZ &x, const ZZ &a, const ZZ &n) {
    ZZ r0 = n, r1 = a % n, t0, t1(1);
    while (!r1.zero()) {
        ZZ q, r; ZZ::divrem(r0, r1, q, r);
        ZZ t2 = t0 - q * t1;
        r0 = r1; r1 = r; t0 = t1; t1 = t2;
    }
    if (!(r0 == ZZ(1))) { x = r0; return 1; }
    x = t0 % n;
    return 0;
}
inline ZZ InvMod(const ZZ &a, const ZZ &n) { ZZ x; if (InvModStatus(x, a, n)) throw std::runtime_error("InvMod: inverse undefined"); return x; }

class ZZ_p {
public:
    ZZ v;                                          // in [0, modulus)
    ZZ_p() {}
    static ZZ &mod() { static thread_local ZZ m(2); return m; }
    static void init(const ZZ &p) { mod() = p; }
    static const ZZ &modulus() { return mod(); }
};
inline const ZZ &rep(const ZZ_p &a) { return a.v; }
inline ZZ_p to_ZZ_p(const ZZ &a) { ZZ_p r; r.v = a % ZZ_p::modulus(); return r; }
inline ZZ_p to_ZZ_p(long a) { return to_ZZ_p(ZZ(a)); }
inline void conv(ZZ_p &x, const ZZ &a) { x = to_ZZ_p(a); }
inline void conv(ZZ_p &x, long a) { x = to_ZZ_p(a); }
inline ZZ_p operator+(const ZZ_p &a, const ZZ_p &b) { ZZ_p r; r.v = a.v + b.v; if (r.v >= ZZ_p::modulus()) r.v -= ZZ_p::modulus(); return r; }
inline ZZ_p operator-(const ZZ_p &a, const ZZ_p &b) { ZZ_p r; r.v = a.v - b.v; if (r.v < ZZ(0)) r.v += ZZ_p::modulus(); return r; }
inline ZZ_p operator-(const ZZ_p &a) { ZZ_p r; if (!a.v.zero()) r.v = ZZ_p::modulus() - a.v; return r; }
inline ZZ_p operator*(const ZZ_p &a, const ZZ_p &b) { ZZ_p r; r.v = (a.v * b.v) % ZZ_p::modulus(); return r; }
inline ZZ_p inv(const ZZ_p &a) { ZZ_p r; r.v = InvMod(a.v, ZZ_p::modulus()); return r; }
inline bool operator==(const ZZ_p &a, const ZZ_p &b) { return a.v == b.v; }
inline bool operator!=(const ZZ_p &a, const ZZ_p &b) { return !(a.v == b.v); }
inline bool operator==(const ZZ_p &a, long b) { return a.v == to_ZZ_p(b).v; }
inline long IsZero(const ZZ_p &a) { return a.v.zero(); }
inline std::ostream &operator<<(std::ostream &os, const ZZ_p &a) { return os << a.v; }

}  // namespace NTL
