// DECLARATIONS ONLY -- never linked.  Used by tests/test_reference_examples_compile.py to run `g++ -fsyntax-only`
// over the reference's example sources against cuhe_amd/cxx/CuHE.h, i.e. to check that every cuHE symbol those
// sources use exists here with a compatible signature.  NTL is not installed in this image; these headers declare
// the part of NTL's modular-polynomial interface the examples mention so that the check can reach the cuHE calls.
#pragma once
#include <NTL/ZZ.h>
#include <NTL/ZZX.h>
namespace NTL {
class ZZ_p { public: ZZ_p(); static void init(const ZZ &p); };
const ZZ &rep(const ZZ_p &a);
class ZZ_pX { public: ZZ_pX(); ZZ_pX &operator=(long); };
ZZ_pX to_ZZ_pX(const ZZX &a);
ZZX to_ZZX(const ZZ_pX &a);
void SetCoeff(ZZ_pX &x, long i, long a);
void SetCoeff(ZZ_pX &x, long i, const ZZ_p &a);
const ZZ_p coeff(const ZZ_pX &a, long i);
long deg(const ZZ_pX &a);
void clear(ZZ_pX &a);
ZZ_pX operator+(const ZZ_pX &, const ZZ_pX &);
ZZ_pX operator-(const ZZ_pX &, const ZZ_pX &);
ZZ_pX operator*(const ZZ_pX &, const ZZ_pX &);
ZZ_pX operator/(const ZZ_pX &, const ZZ_pX &);
ZZ_pX operator%(const ZZ_pX &, const ZZ_pX &);
ZZ_pX &operator%=(ZZ_pX &, const ZZ_pX &);
bool operator==(const ZZ_pX &, const ZZ_pX &);
bool operator!=(const ZZ_pX &, const ZZ_pX &);
void DivRem(ZZ_pX &q, ZZ_pX &r, const ZZ_pX &a, const ZZ_pX &b);
class vec_ZZ_pX { public: void SetLength(long n); long length() const; ZZ_pX &operator[](long i); const ZZ_pX &operator[](long i) const; };
}  // namespace NTL
