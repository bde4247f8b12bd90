// Debug.h -- error convention of the reference (cuhe/Debug.h:35-66): report
// file:line and exit(-1).  Here the checked status is the C-ABI return code.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "../../include/cuhe_hip.h"

#define CSC(status) ::cuHE::detail::safeCall((status), __FILE__, __LINE__)
#define CCE() ((void)0)

namespace cuHE { namespace detail {
// exit(code) -- from one of the gate scheduler's worker threads: the same exit status WITHOUT running the static destructors under the
// feet of the client's threads (a failed allocation inside a recorded gate segfaulted at exit in 3 of 6 injected failures, profiles/
// r06_sched_soak.txt, before this).  Scheduler.cpp.
void die(int code);
inline void safeCall(int status, const char *file, int line) {
	if (status != CUHE_OK) {
		fprintf(stderr, "cuheSafeCall() failed at %s:%i : %s\n", file, line, cuhe_hip_last_error());
		die(-1);
	}
}
}} // namespace cuHE::detail
