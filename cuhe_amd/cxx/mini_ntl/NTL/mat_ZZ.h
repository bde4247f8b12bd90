// mini_ntl -- FALLBACK ONLY (see ZZ.h).  Nothing of this NTL header is used by cuHE clients; present so that their includes resolve.
#pragma once
#include "ZZ_pX.h"
