// Debug.h -- error convention of the reference (cuhe/Debug.h:35-66): report
// file:line and exit(-1).  Here the checked status is the C-ABI return code.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "../../include/cuhe_hip.h"

#define CSC(status) ::cuHE::detail::safeCall((status), __FILE__, __LINE__)
#define CCE() ((void)0)

namespace cuHE { namespace detail {
inline void safeCall(int status, const char *file, int line) {
	if (status != CUHE_OK) {
		fprintf(stderr, "cuheSafeCall() failed at %s:%i : %s\n", file, line, cuhe_hip_last_error());
		exit(-1);
	}
}
}} // namespace cuHE::detail
