// DECLARATIONS ONLY -- see ZZ_pX.h in this directory.  (Nothing of this NTL header is used by the examples.)
#pragma once
#include <NTL/ZZ_pX.h>
