// comm.hpp -- the one exchange of the CRT-prime-sharded multiply + relinearise: an all-gather of CRT rows
// before ICRT, behind the C ABI so that it is enqueued on the caller's compute stream without Python in between.
//
// Two transports:
//  * one process per GPU (bench.py / torchrun, any MPI-style launcher): RCCL.  librccl is opened at run time (dlopen),
//    re-using the copy a host process already loaded (PyTorch ships one), so that libcuhe_hip.so has no link-time
//    dependency on it and single-GPU clients never touch it.  The exchange IS the collective SURVEY 8(e) names: ONE
//    ncclAllGather, in place, when the ranks' blocks are equal (config 4 at level 0: 48 primes over 2 / 4 / 8 ranks); when
//    the number of primes is not a multiple of the number of ranks, one ncclAllGather of blocks padded to the largest
//    (through a staging buffer, unpacked by two strided copies); exchange_path() below is the whole policy.  The group
//    of ncclBroadcast calls of rounds 2-4 (root r sends its block in place) is kept as a selectable fallback.
//  * one process driving several devices (the reference's multiGPUs(n) model, cuhe/CuHE.cu:217-256): peer copies over
//    xGMI ordered by events (cuhe_keyswitch.hip, cuhe_hip_mul_relin_sharded_inproc) -- every block goes straight over the
//    link between its two devices, which is what a direct all-gather of 0.2-1.5 MiB blocks amounts to on a fully
//    connected xGMI topology, and it also runs on the virtual devices the single-GPU tests use.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only; the symbols are resolved with dlsym

namespace cuhe { namespace comm {

struct Api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional (diagnostics: what RCCL itself says about the communicator)
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    const char *error = nullptr;
};

inline Api &api() {
    static Api a;
    if (a.handle || a.error) return a;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) if ((a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // already in the process
    if (!a.handle) for (const char *n : names) if ((a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a.handle) { a.error = "librccl.so not found"; return a; }
#define CUHE_SYM(field, name) *(void **)(&a.field) = dlsym(a.handle, name); if (!a.field) { a.error = "librccl lacks " name; return a; }
    CUHE_SYM(GetUniqueId, "ncclGetUniqueId") CUHE_SYM(CommInitRank, "ncclCommInitRank") CUHE_SYM(CommDestroy, "ncclCommDestroy")
    CUHE_SYM(GroupStart, "ncclGroupStart") CUHE_SYM(GroupEnd, "ncclGroupEnd") CUHE_SYM(Broadcast, "ncclBroadcast")
    CUHE_SYM(AllGather, "ncclAllGather") CUHE_SYM(GetErrorString, "ncclGetErrorString")
#undef CUHE_SYM
    *(void **)(&a.CommCount) = dlsym(a.handle, "ncclCommCount");
    *(void **)(&a.CommUserRank) = dlsym(a.handle, "ncclCommUserRank");
    *(void **)(&a.GetVersion) = dlsym(a.handle, "ncclGetVersion");
    return a;
}

// the symbols api() needs from librccl (tests/test_capi_symbols.py resolves the same list against the librccl of the image)
inline const char *const *required_symbols() {
    static const char *const names[] = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclBroadcast",
                                        "ncclAllGather", "ncclGetErrorString", nullptr};
    return names;
}
// how the CRT rows of a level travel (cuhe_hip_exchange_path reports it without a GPU)
enum Path { kPathNone = 0, kPathAllGather = 1, kPathAllGatherPadded = 2, kPathBroadcastGroup = 3 };
inline const char *path_name(int p) {
    switch (p) {
    case kPathNone: return "one rank: nothing to exchange";
    case kPathAllGather: return "RCCL: one ncclAllGather, in place (equal blocks)";
    case kPathAllGatherPadded: return "RCCL: one ncclAllGather of padded blocks through a staging buffer (unequal blocks)";
    case kPathBroadcastGroup: return "RCCL: group of ncclBroadcast, one per rank's block, in place";
    }
    return "?";
}
// force: 0 / 1 = the policy, 2 = padded all-gather, 3 = broadcast group (tests and A/B runs; CUHE_EXCHANGE=allgather|padded|bcast)
inline int exchange_path(int np, int nranks, int force) {
    if (nranks <= 1 && force <= 0) return kPathNone;
    if (force == 2) return kPathAllGatherPadded;
    if (force == 3) return kPathBroadcastGroup;
    if (np < nranks) return kPathBroadcastGroup;                    // a rank without a block: nothing to pad from
    return np % nranks == 0 ? kPathAllGather : kPathAllGatherPadded;
}
struct State {
    ncclComm_t comm = nullptr; int nranks = 1, rank = 0;
    int force_exchange = 0;               // tests: > 0 exchanges on a communicator of ONE rank too; 2 / 3 also pick the path (exchange_path)
    const char *last_path = "none yet"; long exchanges = 0;
    long path_count[4] = {0, 0, 0, 0};    // exchanges per path (cuhe_hip_comm_info)
    unsigned *stage = nullptr; size_t stage_words = 0; int stage_dev = -1;      // staging buffer of the padded all-gather
    void *stage_event = nullptr, *stage_stream = nullptr; bool stage_used = false;   // its last use: the event recorded behind the unpack copies, and on which stream
};
inline State &state() { static State s; return s; }

// contiguous, balanced blocks: the first (np % nranks) ranks own one prime more (== cuhe_amd/sharded.py: shard_bounds)
inline void shard_bounds(int np, int nranks, int rank, int *first, int *count) {
    const int base = np / nranks, extra = np % nranks;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + (rank < extra ? rank : extra);
}

}}  // namespace cuhe::comm
