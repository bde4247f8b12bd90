// DECLARATIONS ONLY -- see ZZ_pX.h in this directory.
#pragma once
#include <NTL/ZZ_pX.h>
namespace NTL {
class ZZ_pE { public: ZZ_pE(); static void init(const ZZ_pX &modulus); };
ZZ_pE to_ZZ_pE(const ZZ_pX &a);
ZZ_pE inv(const ZZ_pE &a);
const ZZ_pX &rep(const ZZ_pE &a);
}  // namespace NTL
